import os, sys
sys.path.insert(0, os.getcwd())
from fugue_amd import engine as E, workloads as W
cp = E.compile_model(W.normal_sites(2))
eng = E.Engine(cp, 64, seed=1)
uid = E.comm_unique_id()
print("uid ok", len(uid), flush=True)
try:
    comm = eng.comm_init(1, 0, uid)
    print("comm", comm, flush=True)
    E.comm_destroy(comm)
except Exception as ex:
    print("FAIL", ex, flush=True)
