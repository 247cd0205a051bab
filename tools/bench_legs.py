"""Print the legs of a bench.py JSON line: `python tools/bench_legs.py <file>`"""
import json, sys
p = json.load(open(sys.argv[1]))
print("headline", round(p["value"] / 1e9, 3), "G patches/s, kernel frac", p["roofline"]["frac"], "avg_us", p["roofline"]["avg_us"])
for k in ("with_attn", "configs[1]", "single_slide", "slide_sized_bags", "eval_loop_lookahead", "eval_loop_lookahead_slide_sized", "strong_scaling_base"):
    v = p.get(k)
    if v:
        print(f"  {k:34s} us/bag {v.get('us_per_bag')}  whole {v.get('whole_step_frac_of_hbm_roofline', v.get('frac_of_hbm_roofline'))}  kernel {v.get('kernel', {}).get('frac')}  ok {v.get('verified', {}).get('ok')}")
