"""Dev tool (VERDICT r03 task 4): does the 256 MiB Infinity Cache (MALL) hold a producer's output for the consumer that runs next?

gfx950 / ROCm 7.2 expose no MALL hit counter (`rocprofv3 -L`: profiles/r04_counters_tcc.txt), so the cache is measured in TIME: a
consumer pass over a tensor T of S bytes is timed alone (HIP events around that one launch)
  hot  : immediately after the producer launch that WROTE T (what a train step does: BN apply right after the convolution, the
         dgrad right after the BN-backward apply, ...);
  warm : T was READ (not written) by the previous launch;
  cold : after a 1.2 GB flush copy ran between producer and consumer (nothing of T can be left in L2 / MALL).
Consumers: cvhip_copy2d (1 read + 1 write), cvhip_bn_act_fwd SiLU (1 read + 1 write), cvhip_bn_act_bwd_partial (2 reads).
hot / cold is what producer->consumer adjacency is worth; when it is ~1 the consumer is not limited by where its operand lives."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
REPS = int(os.environ.get("REPS", "12"))
flush_a = torch.empty(600 * 1024 * 1024 // 2, dtype=torch.bfloat16, device=dev)
flush_b = torch.empty_like(flush_a)


def flush():
    flush_b.copy_(flush_a)


def timed(pre, fn):
    """median over REPS of: pre() (untimed, same stream) then fn() between two events"""
    ts = []
    for _ in range(REPS):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


print("%-34s %9s %9s %9s   %s" % ("consumer, tensor", "hot us", "warm us", "cold us", "hot/cold   TB/s hot / cold"))
for (M, C) in [(64 * 40 * 40, 128), (64 * 40 * 40, 256), (64 * 80 * 80, 128), (64 * 160 * 160, 64), (64 * 320 * 320, 32)]:
    E = M * C * 2
    src = torch.randn(M, C, device=dev).to(torch.bfloat16)
    T = torch.empty_like(src)
    T2 = torch.randn(M, C, device=dev).to(torch.bfloat16)
    out = torch.empty_like(src)
    sc = torch.rand(C, device=dev) + 0.5
    sh = torch.randn(C, device=dev)
    mean = torch.randn(C, device=dev) * 0.1
    inv = torch.rand(C, device=dev) + 0.5
    rows = L.load().cvhip_colreduce_rows(M, C)
    partial = torch.empty(rows + L.REDUCE_SCRATCH_ROWS, 2, C, device=dev)
    produce = lambda: L.call("cvhip_copy2d", src.data_ptr(), C, T.data_ptr(), C, M, C, st)
    produce2 = lambda: (L.call("cvhip_copy2d", src.data_ptr(), C, T.data_ptr(), C, M, C, st), L.call("cvhip_copy2d", src.data_ptr(), C, T2.data_ptr(), C, M, C, st))
    readonly = lambda: L.call("cvhip_bn_act_bwd_partial", T.data_ptr(), C, T.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), L.ACT_SILU, 0.0, partial.data_ptr(), st)
    cons = {
        "copy2d (r+w)": (2 * E, produce, lambda: L.call("cvhip_copy2d", T.data_ptr(), C, out.data_ptr(), C, M, C, st)),
        "bn_act_fwd silu (r+w)": (2 * E, produce, lambda: L.call("cvhip_bn_act_fwd", T.data_ptr(), C, out.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), L.ACT_SILU, 0.0, None, 0, st)),
        "bn_act_bwd_partial silu (2r)": (2 * E, produce2, lambda: L.call("cvhip_bn_act_bwd_partial", T.data_ptr(), C, T2.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), L.ACT_SILU, 0.0, partial.data_ptr(), st)),
    }
    for name, (nbytes, prod, fn) in cons.items():
        hot = timed(lambda: (flush(), prod()), fn)
        warm = timed(lambda: (prod(), flush(), readonly() if "2r" not in name else (readonly(), L.call("cvhip_bn_act_bwd_partial", T2.data_ptr(), C, T2.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), L.ACT_SILU, 0.0, partial.data_ptr(), st))), fn)
        cold = timed(lambda: (prod(), flush()), fn)
        print("%-34s %9.1f %9.1f %9.1f   %5.2f     %5.2f / %5.2f" % ("%s %d MB" % (name, E // 1000000), hot, warm, cold, hot / cold, nbytes / hot / 1e6, nbytes / cold / 1e6))
    del src, T, T2, out
    torch.cuda.empty_cache()
