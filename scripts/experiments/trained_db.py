#!/usr/bin/env python3
"""Does the pipeline RECOGNISE places when the database is trained on them?  The benchmark's 100 000-place database is
the bundled model's rows permuted (throughput only, Recall at chance).  Here P synthetic places each have a rate image
(Gamma(0.5) * 6 per pixel); the reference traverse is one Poisson draw per place, the query traverse another.  Both
layers are trained on the reference traverse with the GPU trainer (k_train_layer, the reference's schedule), the
queries run through the inference path, and Recall@K against "query i is place i" is printed for the raw similarity
matrix (argmax per query) and after the sequence matcher.

usage: trained_db.py [P] [epochs] [carry 0|1]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    carry = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    import main as cli
    from lens_amd.engine import LensEngine, seqmatch_topk
    from lens_amd.synthetic import synthetic_traverses, train_on_traverse
    roi, dims, T = 128, 10, 250
    ref, qry = synthetic_traverses(P, roi)
    a = cli.build_parser().parse_args(["--train_model", "--reference_places", str(P), "--query_places", str(P),
                                       "--epoch_feat", str(epochs), "--epoch_out", str(epochs), "--roi_dim", str(roi)])
    t0 = time.time()
    w1, w2 = train_on_traverse(a, ref)
    torch.cuda.synchronize()
    print(f"trained {P} places, {epochs} epochs per layer: {time.time() - t0:.2f} s", flush=True)
    eng = LensEngine(w1, w2, roi, dims, T, device="cuda", max_batch=min(P, 4096))
    rows = []
    for c0 in range(0, P, 4096):
        rows.append(eng.forward_counts(qry[c0:c0 + 4096].cuda(), carry=carry).clone())
    S = torch.cat(rows, 0)
    truth = torch.arange(P, device=S.device)
    top = S.topk(5, dim=1).indices
    print(f"similarity matrix: recall@1 {(top[:, 0] == truth).float().mean().item():.4f}  "
          f"recall@5 {(top == truth[:, None]).any(1).float().mean().item():.4f}  (chance {1 / P:.5f})")
    idx, _ = seqmatch_topk(S, 2, 5)
    n = idx.shape[0]
    print(f"sequence matcher L=2: recall@1 {(idx[:, 0] == truth[:n]).float().mean().item():.4f}  "
          f"recall@5 {(idx == truth[:n, None]).any(1).float().mean().item():.4f}")
    # the same queries against themselves as references (an upper bound: no Poisson noise between the traverses)
    S0 = torch.cat([eng.forward_counts(ref[c0:c0 + 4096].cuda(), carry=carry).clone() for c0 in range(0, P, 4096)], 0)
    print(f"reference traverse as queries: recall@1 {(S0.argmax(1) == truth).float().mean().item():.4f}")


if __name__ == "__main__":
    main()
