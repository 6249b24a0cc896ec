#!/usr/bin/env python3
"""Two processes, one communicator of the C library (rt_hip_comm_init / rt_hip_render_ranks), one GPU each:

  python3 tools/comm_two_ranks.py [--devices 0,1]

With both ranks on ONE device (--devices 0,0) RCCL refuses the communicator; the tool then reports how (it must be an
error code and a message, not a hang)."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
from raytracer_amd import parallel, abi
from helpers import c1
rank, world, device, idfile = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
if rank == 0:
    cid = parallel.RankComm.new_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(cid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            sys.exit("no id file")
        time.sleep(0.05)
    cid = open(idfile, "rb").read()
ri = c1(width=40, height=33, batch=8, max_batches=5, min_batches=1)
res = ri.resident(device)
try:
    comm = parallel.RankComm(cid, rank, world, device)
except abi.RtError as e:
    print("rank", rank, "COMM-REFUSED", e, flush=True)
    sys.exit(3)
for block in (1, 8):
    cfg = ri.config.copy()
    cfg.row_block = block
    out, st = comm.render(res, cfg)
    if rank == 0:
        one, st1 = res.render(ri.config.copy())
        assert np.array_equal(out["pixels"], one["pixels"]) and np.array_equal(out["sample_pixels"], one["sample_pixels"])
        assert st.rays == st1.rays
print("rank", rank, "OK", flush=True)
comm.close()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0,1")
    args = ap.parse_args()
    devices = [int(d) for d in args.devices.split(",")]
    d = tempfile.mkdtemp()
    script = os.path.join(d, "worker.py")
    with open(script, "w") as f:
        f.write(WORKER)
    idfile = os.path.join(d, "comm_id")
    procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), str(len(devices)), str(dev), idfile], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r, dev in enumerate(devices)]
    t0 = time.time()
    rcs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1, 90 - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\nTIMEOUT"
        rcs.append(p.returncode)
        print(out[-1500:])
    print("return codes", rcs)
    sys.exit(0 if all(rc == 0 for rc in rcs) else 1)


if __name__ == "__main__":
    main()
