"""One rank of the world > 1 tests of dcp_unwarp_stack_rows_rccl_f32 (tests/test_rccl_world.py starts N of these on the ONE GPU
of the test box, DCP_RCCL_PATH pointing at tests/c/libfake_rccl.so).  No torch, no oracle in here: the rank writes what it
received to <outdir>/rank<r>.npy and the parent compares every rank's copy with the oracle."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from discorpy_amd import _ffi as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--idfile", required=True)
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--counts", required=True, help="projections per rank, comma separated")
    ap.add_argument("--pipelines", default="1", help="comma separated; one exchange per entry, results rank<r>_p<pipeline>.npy")
    ap.add_argument("--shape", default="300,520")
    ap.add_argument("--rows", default="100,64", help="row_start,nrows")
    ap.add_argument("--seed", type=int, default=77)
    ap.add_argument("--own-stream", action="store_true")
    ap.add_argument("--fixed-repeats", type=int, default=0,
                    help="after the exchanges: dcp_rccl_comm_fixed_shards(1), then the last exchange this many times back to back "
                         "(no synchronisation in between), result rank<r>_fixed.npy; then every rank breaks the promise the same way")
    ap.add_argument("--disagree", default=None, choices=[None, "nrows", "null", "pipeline"],
                    help="the LAST rank passes a different nrows / a null volume / another pipeline: every rank must get an error")
    a = ap.parse_args()
    L = F.lib()
    F.require_device()
    assert os.environ.get("DCP_RCCL_PATH"), "the tests run this against the stand-in only"
    counts = [int(v) for v in a.counts.split(",")]
    assert len(counts) == a.world
    H, W = (int(v) for v in a.shape.split(","))
    row0, nrows = float(a.rows.split(",")[0]), int(a.rows.split(",")[1])
    D = sum(counts)
    d0 = sum(counts[:a.rank])
    dl = counts[a.rank]
    whole = np.random.default_rng(a.seed).random((max(D, 1), H, W), dtype=np.float32)[:D]
    xc, yc, fact = 250.3, 160.7, [1.0, -2e-5, 3e-8]
    fa, nf = F.fact_array(fact)
    idbuf = (C.c_char * 128)()
    if a.rank == 0:
        F.check(L.dcp_rccl_unique_id(idbuf, 128))
        open(a.idfile + ".tmp", "wb").write(bytes(idbuf))
        os.replace(a.idfile + ".tmp", a.idfile)
    else:
        t0 = time.time()
        while not os.path.exists(a.idfile):
            if time.time() - t0 > 120:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.02)
        idbuf = (C.c_char * 128).from_buffer_copy(open(a.idfile, "rb").read())
    comm = C.c_void_p()
    F.check(L.dcp_rccl_comm_create(C.byref(comm), a.world, a.rank, idbuf, 0))
    dvol = F.DeviceBuffer(max(dl * H * W * 4, 4))
    if dl:
        dvol.upload(np.ascontiguousarray(whole[d0:d0 + dl]))
    dout = F.DeviceBuffer(max(D * nrows * W * 4, 4))
    stream = C.c_void_p()
    if a.own_stream:
        F.check(L.dcp_stream_create(C.byref(stream), 0))
    report = {"rank": a.rank, "rc": [], "err": []}
    last = a.rank == a.world - 1
    for p in [int(v) for v in a.pipelines.split(",")]:
        dout.upload(np.full((max(D, 1), nrows, W), np.nan, np.float32)[:D] if D else np.zeros(1, np.float32))
        F.check(L.dcp_stream_synchronize(0, None))
        call_rows = nrows - 1 if (a.disagree == "nrows" and last) else nrows
        call_vol = None if (a.disagree == "null" and last) else dvol.ptr
        call_pipe = p + 1 if (a.disagree == "pipeline" and last) else p
        rc = L.dcp_unwarp_stack_rows_rccl_f32(call_vol, dout.ptr, dl, H, W, H * W, W, xc, yc, fa, nf, row0, call_rows, 1, F.BLEND_F64LERP, comm,
                                              call_pipe, stream)
        report["rc"].append(int(rc))
        report["err"].append(F.last_error() if rc else "")
        F.check(L.dcp_stream_synchronize(0, stream))
        if rc == 0 and D:
            np.save(os.path.join(a.outdir, "rank%d_p%d.npy" % (a.rank, p)), dout.download((D, nrows, W), np.float32))
    def info_of():
        info, depths, path = (C.c_int64 * 10)(), (C.c_int64 * a.world)(), C.create_string_buffer(1024)
        F.check(L.dcp_rccl_comm_info(comm, info, 10, depths, a.world, path, 1024))
        return {"info": [int(v) for v in info], "shard_depths": [int(v) for v in depths], "librccl": path.value.decode()}
    report["comm"] = info_of()
    if a.fixed_repeats > 0 and a.disagree is None:
        p = int(a.pipelines.split(",")[-1])
        F.check(L.dcp_rccl_comm_fixed_shards(comm, 1))
        dout.upload(np.full((max(D, 1), nrows, W), np.nan, np.float32)[:D] if D else np.zeros(1, np.float32))
        F.check(L.dcp_stream_synchronize(0, None))
        for _ in range(a.fixed_repeats):
            F.check(L.dcp_unwarp_stack_rows_rccl_f32(dvol.ptr, dout.ptr, dl, H, W, H * W, W, xc, yc, fa, nf, row0, nrows, 1, F.BLEND_F64LERP, comm,
                                                     p, stream))
        F.check(L.dcp_stream_synchronize(0, stream))
        if D:
            np.save(os.path.join(a.outdir, "rank%d_fixed.npy" % a.rank), dout.download((D, nrows, W), np.float32))
        report["comm_after_fixed"] = info_of()
        # every rank passes one row fewer: each is refused locally, nobody enters a collective
        report["rc_broken_promise"] = int(L.dcp_unwarp_stack_rows_rccl_f32(dvol.ptr, dout.ptr, dl, H, W, H * W, W, xc, yc, fa, nf, row0, nrows - 1, 1,
                                                                           F.BLEND_F64LERP, comm, p, stream))
        report["err_broken_promise"] = F.last_error()
    F.check(L.dcp_rccl_comm_destroy(comm))
    json.dump(report, open(os.path.join(a.outdir, "rank%d.json" % a.rank), "w"))


if __name__ == "__main__":
    main()
