"""How many evaluate_synset epochs the bench's eval leg needs on the template pool before top-1 says something (one-off probe)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for noise, ep in ((1.0, 400), (2.0, 500), (3.0, 500), (1.0, 300)):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--sustain-seconds", "0",
                          "--no-extra-legs", "--eval-epochs", str(ep), "--pool-noise", str(noise)], cwd=root, capture_output=True, text=True)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print(noise, ep, d["eval"])
