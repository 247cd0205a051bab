#!/bin/bash
# rocprofv3 kernel stats of a command, summary copied next to the scratch: tools/prof.sh <tag> <cmd...>   (ON THE GPU BOX, via gpurun)
tag=$1; shift
O=gpurun_out/${VLSA_ROUND:-r06}
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- "$@" > $O/prof_$tag.out 2>&1
f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $O/${tag}_kernel_stats.csv
rm -rf $O/prof_$tag
python tools/kstats.py $O/${tag}_kernel_stats.csv | head -${PROF_HEAD:-25}
