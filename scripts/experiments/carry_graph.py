"""How much of the pipelined carried stream's time is launch overhead?  The same 256-query call captured in a HIP graph
(torch.cuda.CUDAGraph: both HIP streams of the pipeline, their events) and replayed, against the eager call.
Every replay is checked against the eager result.  python carry_graph.py [P]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from lens_amd.engine import LensEngine
from lens_amd.synthetic import synthetic_frames, synthetic_database
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
w1, w2p = bench.pretrained()
Q = 256
eng = LensEngine(w1, synthetic_database(w2p, P), bench.ROI, bench.DIMS, bench.T_STEPS, device="cuda", max_batch=Q)
fr = synthetic_frames(Q, bench.ROI, seed=1234).cuda()
def timed(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eng.reset_state(); want = eng.forward_counts(fr, carry=True).clone(); v_want = eng.v_out.clone()
def eager():
    eng.reset_state_inplace() if hasattr(eng, "reset_state_inplace") else (eng.v_feat.zero_(), eng.v_out.zero_())
    return eng.forward_counts(fr, carry=True)
print("eager:", timed(eager), "ms per 256 queries")
out = torch.empty((Q, eng.P), device="cuda")
eng.v_feat.zero_(); eng.v_out.zero_()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng._carried_stream_pipelined(fr, out)           # warm-up on the capture stream (workspaces, side stream)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
eng.v_feat.zero_(); eng.v_out.zero_()
with torch.cuda.graph(g, stream=s):
    eng._carried_stream_pipelined(fr, out)
def replay():
    eng.v_feat.zero_(); eng.v_out.zero_()
    g.replay()
for i in range(3):
    replay(); torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(eng.v_out, v_want), f"replay {i} differs"
    torch.randn(1000, device="cuda").sum().item()    # eager work between the replays
print("graph replay:", timed(replay), "ms per 256 queries (3 replays checked against the eager result)")
