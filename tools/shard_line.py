import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("stream mode", sys.argv[1], "frames", d["config"]["frames"], "ms/step", round(d["ms_per_step"], 1), "us/iter",
      round(d["ms_per_step"] * 2, 1), "in-loop NN ms", round(d["roofline"]["ms_per_launch"], 3))
