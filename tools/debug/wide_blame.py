#!/usr/bin/env python3
"""Debug: WHAT went wrong in a wrong block of the wide GEMM.  Runs the form under test (the environment's SYLDET_LIB /
SYLDET_WIDE_* switches) N times at a size that takes many rounds of workgroups, compares every run with the shipped order (made
in the same process under SYLDET_WIDE_NOSTAGGER=1), and for the first wrong blocks of 16 evaluations rebuilds the chunk
arithmetic on the host (bf16 operands from the |X| columns of syldet_spectrogram, the folded tables of upload_wide restated in
numpy, float64 sums) and scores a list of single-fault hypotheses against the observed difference:

    late(c,ut,s)    the hidden values of chunk c, unit tile ut were made from accumulators that lacked k-steps >= s
                    (the epilogue read them before the matrix pipe wrote them)
    skip(c,ut,s)    k-step s of (c, ut) missing from the sums
    ashift(c,ut,s,d) k-step s multiplied the fragment of k-step s + d
    aother(c,ut,s,c') k-step s multiplied chunk c''s fragment (a stale or early LDS buffer)
    afrom(c,ut,s,c') k-steps >= s multiplied chunk c''s fragments
    bias(c,ut,c')   the accumulators started at chunk c''s bias
    w1(c,ut,c')     the second-layer weights of chunk c' met the hidden values of chunk c
    prev(c,ut)      the hidden values of chunk c - 1 met the second-layer weights of chunk c (accumulators not yet replaced)
    drop(c,ut) / dup(c,ut)   the tile's share of the output missing / added twice
    bt1(c,ut,s)     k-step s multiplied the OTHER evaluation tile's operands

and prints the best few with their residuals (|predicted - observed| / |observed| over the block's 16 evaluations).

    [SYLDET_LIB=...] python tools/debug/wide_blame.py [--runs 8] [--blocks 12] [--C 64] [--S 8388608]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=8)
ap.add_argument("--blocks", type=int, default=12)
ap.add_argument("--C", type=int, default=64)
ap.add_argument("--S", type=int, default=1 << 23)
ap.add_argument("--tile", type=int, default=256, help="evaluations per workgroup (256: two workgroups of 8 waves a CU)")
a = ap.parse_args()

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wide_model import cfg, I, F, T, L0, L1, H
x = synth.channels_on_device(a.C, a.S, torch.device("cuda", 0), fs=cfg.samplingRate)


from wide_model import bf16, Wq, bias, w1q, NCH, r_of
import wide_model

os.environ["SYLDET_WIDE_NOSTAGGER"] = "1"            # the reference: round 4's order, every wave of a workgroup in the same chunk
with sd.SyllableDetector(cfg, channels=a.C, engine=_abi.ENGINE_WIDE_BF16) as det_ref:
    ref = det_ref.run(x)[0][..., 0].cpu().numpy().copy()
    again = det_ref.run(x)[0][..., 0].cpu().numpy()
    assert (again == ref).all(), "the shipped order itself differs run to run"
    cols = det_ref.spectrogram(x)
    torch.cuda.synchronize()
os.environ.pop("SYLDET_WIDE_NOSTAGGER")


def blame(c, e0, obs):
    """obs[16] = wrong - right of evaluations e0 .. e0 + 15 (an evaluation tile of one wave)"""
    t = (e0 // 16) & 1
    base = e0 - 16 * t
    xb32 = wide_model.operands(cols, c, base, 32)
    xb, xo = xb32[16 * t:16 * t + 16], xb32[16 * (1 - t):16 * (1 - t) + 16]         # this tile's operands, the other tile's
    P = np.einsum("uks,nks->nus", Wq.reshape(H, 10, 32).transpose(0, 2, 1), xb.reshape(16, 10, 32).transpose(0, 2, 1))  # [16][H][10]
    acc = bias[None, :] + P.sum(-1)
    good = w1q[None, :] * r_of(acc)                                                  # [16][H] every unit's share of the output
    emu = good.sum(1) + float(L1.biases[0]) + float(L1.weights.astype(np.float64).sum())
    print("      (host model against the shipped order on this block: max |difference| %.3g)" % float(np.abs(emu - ref[c, e0:e0 + 16]).max()))
    cand = []

    def tile(v):                                                                     # [16][H] -> [16][NCH][2]
        return v.reshape(16, NCH, 2, 16).sum(-1)

    def add(name, pred_tiles):                                                       # pred_tiles [16][NCH][2]: predicted change of the output
        pred_tiles = np.concatenate([pred_tiles, pred_tiles.sum(-1, keepdims=True)], -1)   # (unit tile 2 = both tiles of the chunk)
        res = np.linalg.norm(pred_tiles - obs[:, None, None], axis=0) / np.linalg.norm(obs)
        for cc, ut in zip(*np.unravel_index(np.argsort(res, axis=None)[:2], res.shape)):
            cand.append((float(res[cc, ut]), name, int(cc), int(ut)))

    csum = np.cumsum(P, -1)
    for s in range(0, 10):                                                           # late: k-steps < s only
        part = bias[None, :] + (csum[..., s - 1] if s > 0 else 0.0)
        add("late(s=%d)" % s, tile(w1q[None, :] * r_of(part) - good))
        add("skip(s=%d)" % s, tile(w1q[None, :] * r_of(acc - P[..., s]) - good))
        Po = np.einsum("uk,nk->nu", Wq[:, 32 * s:32 * s + 32], xo[:, 32 * s:32 * s + 32])
        add("bt1(s=%d)" % s, tile(w1q[None, :] * r_of(acc - P[..., s] + Po) - good))
        for dlt in (-1, 1):
            if 0 <= s + dlt < 10:
                Ps = np.einsum("uk,nk->nu", Wq[:, 32 * (s + dlt):32 * (s + dlt) + 32], xb[:, 32 * s:32 * s + 32])
                add("ashift(s=%d,d=%+d)" % (s, dlt), tile(w1q[None, :] * r_of(acc - P[..., s] + Ps) - good))
    for dc in (-3, -2, -1, 1, 2, 3):                                                 # another chunk's bytes in this chunk's place
        Wo = np.roll(Wq.reshape(NCH, 32, 320), -dc, axis=0).reshape(H, 320)          # chunk c + dc where chunk c should be
        Pq = np.einsum("uks,nks->nus", Wo.reshape(H, 10, 32).transpose(0, 2, 1), xb.reshape(16, 10, 32).transpose(0, 2, 1))
        for s in range(10):
            add("aother(s=%d,c%+d)" % (s, dc), tile(w1q[None, :] * r_of(acc - P[..., s] + Pq[..., s]) - good))
            add("afrom(s=%d,c%+d)" % (s, dc), tile(w1q[None, :] * r_of(acc - P[..., s:].sum(-1) + Pq[..., s:].sum(-1)) - good))
        bo = np.roll(bias.reshape(NCH, 32), -dc, axis=0).reshape(H)
        add("bias(c%+d)" % dc, tile(w1q[None, :] * r_of(acc - bias[None, :] + bo[None, :]) - good))
        wo = np.roll(w1q.reshape(NCH, 32), -dc, axis=0).reshape(H)
        add("w1(c%+d)" % dc, tile(wo[None, :] * r_of(acc) - good))
    accp = np.roll(acc.reshape(16, NCH, 32), 1, axis=1).reshape(16, H)               # chunk c - 1's accumulators in chunk c's place
    add("prev", tile(w1q[None, :] * r_of(accp) - good))
    add("drop", tile(-good))
    add("dup", tile(good))
    cand.sort()
    return cand[:4]


blocks_done = 0
with sd.SyllableDetector(cfg, channels=a.C, engine=_abi.ENGINE_WIDE_BF16) as det:
    det.profile(True)
    for k in range(a.runs):
        o = det.run(x)[0][..., 0]
        torch.cuda.synchronize()
        if k == 0:
            print("kernels under test:", [n for n, _ in det.lastTimings()], "library:", os.environ.get("SYLDET_LIB", "(the tree's)"), flush=True)
        o = o.cpu().numpy()
        d = np.argwhere(o != ref)
        blocks = sorted({(int(c), int(e) // 16 * 16) for c, e in d})
        print("run %d: %d evaluations in %d blocks differ from the shipped order" % (k, len(d), len(blocks)), flush=True)
        for c, e0 in blocks:
            if blocks_done >= a.blocks:
                break
            blocks_done += 1
            obs = (o[c, e0:e0 + 16].astype(np.float64) - ref[c, e0:e0 + 16].astype(np.float64))
            wave, t = (e0 % a.tile) // 32, (e0 // 16) & 1
            print("  channel %d evaluations %d..%d (workgroup %d, wave %d, tile %d), |diff| %.3g:" % (c, e0, e0 + 15, e0 // a.tile, wave, t, float(np.abs(obs).max())))
            best = blame(c, e0, obs)
            for res, name, cc, ut in best:
                print("      residual %.4f  %-22s chunk %3d unit tile %s" % (res, name, cc, "both" if ut == 2 else str(ut)), flush=True)
print("blamed %d blocks" % blocks_done)
