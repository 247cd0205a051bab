"""The bench.py `train_step` leg alone (BASELINE configs[4]): `python tools/bench_train_step.py [tcga|50k] [steps]`.
VLSA_BENCH_TRAIN_MODE=eager|graph|auto, VLSA_BENCH_ADAM_FUSED=0|1 as in bench.py."""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "tcga"
steps = sys.argv[2] if len(sys.argv) > 2 else "40"
env = dict(os.environ, VLSA_BENCH_ONLY_TRAIN=which)
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "8", "--no-cpu-baseline"], env=env,
                     capture_output=True, text=True)
sys.stderr.write(out.stderr[-6000:])
sys.stderr.write(f"\n[bench_train_step] bench.py exit code {out.returncode}\n")
line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "{}"
try:
    d = json.loads(line)
    print(json.dumps(d.get("train_step", d), indent=1))
except Exception:
    print(out.stdout[-3000:])
sys.exit(out.returncode)
