"""How does the CPU baseline (oracle actor processes) scale on this host?  Prints moves/s for several process counts."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import baseline

if __name__ == "__main__":
    out = []
    for cores in [int(x) for x in sys.argv[1:]] or [8, 32, 64, 128, 256]:
        r = baseline.run(cores, seconds=12.0)
        out.append(dict(cores=cores, total=round(r["value"], 2), per_core=round(r["per_core"], 4)))
        print(json.dumps(out[-1]), flush=True)
