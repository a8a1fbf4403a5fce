"""Power and clock of the GPU while bench.py runs: is the train step power-limited?

Starts `python bench.py --steps N --no-cpu-baseline` as a child process and polls `rocm-smi` (average socket power,
sclk / mclk) every ~0.25 s; prints the idle reading, the readings during the timed region and the board's power cap.

    python benchmarks/power_trace.py [steps] [file for the bench's JSON line]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smi(*args):
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--json", *args], capture_output=True, text=True, timeout=10).stdout
        return json.loads(out[out.index("{"):])
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def reading():
    d = smi("--showpower", "--showclocks")
    card = next(iter(d.values())) if d and "error" not in d else {}
    power = next((v for k, v in card.items() if "ower" in k and "W" in k), None)
    sclk = next((v for k, v in card.items() if k.startswith("sclk")), None)
    mclk = next((v for k, v in card.items() if k.startswith("mclk")), None)
    return power, sclk, mclk, (d.get("error") if isinstance(d, dict) else None)


def main():
    steps = sys.argv[1] if len(sys.argv) > 1 else "300"
    json_out = sys.argv[2] if len(sys.argv) > 2 else None          # where the bench's full JSON line goes
    print("power cap:", json.dumps(smi("--showmaxpower")))
    print("idle:", reading())
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "10",
                              "--no-cpu-baseline", "--no-variants"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    t0 = time.time()
    rows = []
    while child.poll() is None:
        rows.append((time.time() - t0, *reading()[:3]))
        time.sleep(0.25)
    out = child.stdout.read().strip().splitlines()
    print(out[-1][:400] if out else "no bench output")
    if out and json_out:
        with open(json_out, "w") as fh:
            fh.write(out[-1] + "\n")
        d = json.loads(out[-1])
        print("ms per step, by 100 steps:", d.get("ms_per_step_by_100_steps"))
    for t, p, s, m in rows:
        print(f"  t={t:6.2f} s  power {p}  sclk {s}  mclk {m}")


if __name__ == "__main__":
    main()
