"""Host-side bookkeeping that the GPU legs rely on, checked on the CPU: the kernel symbols bench.py rebuilds for its roofline entries
are the names rocprofv3 lists for the same launches (committed summary), its traffic lookup finds the dominant kernel's PMC entry, and
the staging-block layout of upload_batch."""
import csv
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats_names():
    with open(os.path.join(ROOT, "profiles", "r06_kernel_stats.csv")) as fh:
        return {r["Name"] for r in csv.DictReader(fh)}


def test_rebuilt_gemm_symbols_are_the_names_rocprof_lists():
    import bench
    names = _stats_names()
    base = dict(transA=0, transB=1, rowscale=False, bf16=False, tile=0)
    recs = [dict(base, h2=True, h2w=True, epi=3), dict(base, h2=True, h2w=True, epi=2), dict(base, h2=True, epi=6, transA=1, transB=0),      # h2w: the NT forms on the 64-byte-piece kernel
            dict(base, dmf=True, h2out=True, f16p=True, epi=0),          # -> k_dm_mulpred_fused<3>
            dict(base, x3=True, tile=1, epi=1, transA=0, transB=0, rowscale=True),          # scorer layer 1 forward (row-scale prologue), six bf16 products
            dict(base, x3=True, x2h=True, tile=0, epi=6, transA=1, transB=0, rowscale=True),          # its weight gradient, two fp16 planes
            dict(base, x3=True, tile=0, epi=6, transA=1, transB=0),                                   # bf16x3 (the other wide weight gradients)
            dict(base, tile=5, epi=6, transA=1, transB=0, M=64, N=32), dict(base, tile=5, epi=6, transA=1, transB=0, M=128, N=64)]      # small-output TN kernel (round 6)
    for r in recs:
        sym = bench.gemm_symbol(r)
        assert sym in names, "%s is not a kernel of profiles/r06_kernel_stats.csv" % sym


def test_committed_traffic_file_has_the_dominant_kernels():
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench.json")).read())
    dom = line["roofline"]["kernel"].split(" = ")[0]
    assert dom in d and d[dom]["traffic_bytes_per_launch"] > line["roofline"]["algorithmic_bytes_per_launch"] * 0.9
    for g in line["roofline"]["top_gemms"]:
        assert g["kernel"].split(" = ")[0] in d


def test_pack_offsets_layout():
    from chameleon_recsys_amd.nar.nar_model import pack_offsets
    arrs = dict(a=np.zeros((3, 5), np.int64), b=np.zeros(7, np.uint8), c=np.zeros((2, 0), np.float32), d=np.zeros(65, np.int32))
    offs, total = pack_offsets(arrs)
    assert offs == dict(a=0, b=256, c=512, d=512) and total == 1024
    for k, a in arrs.items():
        assert offs[k] % 256 == 0 and offs[k] + a.nbytes <= total


def test_no_collective_under_a_rank0_branch_of_bench():
    """bench.py under torchrun: a collective that only rank 0 reaches deadlocks or - once the other ranks have left - fails with
    'connection closed by peer' (round 3's final barrier sat inside `if rank == 0`; no N > 1 run existed to catch it)."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    collectives = {"barrier", "all_reduce", "broadcast", "all_gather", "reduce_scatter_tensor", "all_gather_into_tensor", "reduce", "gather"}
    bad = []

    def walk(node, under_rank0):
        if isinstance(node, ast.If):
            r0 = ast.unparse(node.test).replace(" ", "") == "rank==0"
            for n in node.body:
                walk(n, under_rank0 or r0)
            for n in node.orelse:
                walk(n, under_rank0)
            return
        if isinstance(node, ast.Call) and under_rank0:
            f = ast.unparse(node.func)
            if f.startswith("dist.") and f.split(".")[1] in collectives:
                bad.append((node.lineno, f))
        for n in ast.iter_child_nodes(node):
            walk(n, under_rank0)
    walk(tree, False)
    assert not bad, bad
