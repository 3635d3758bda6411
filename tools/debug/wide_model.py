"""Debug: the wide GEMM's arithmetic restated on the host for tools/debug/wide_blame.py and wide_trace.py -- upload_wide's
folded tables (csrc/syldet_api.cpp) in numpy, the kernel front's bf16 operands from the |X| columns, float64 sums."""
import numpy as np
from syllable_detector_swift_amd import nets

cfg = nets.wide_mlp(nets.from_npz())
I, F = cfg.net.inputs, 29
T = I // F
L0, L1 = cfg.net.layers
H = L0.outputs


def bf16(v):
    """float32 array -> float32 array holding the round-to-nearest-even bf16 values (upload_wide's to_bf16, v_cvt_pk_bf16_f32)"""
    u = np.ascontiguousarray(v, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


# upload_wide (csrc/syldet_api.cpp) restated: the affine input maps folded into the first layer, tanh folded into the tables
fa, fo = np.ones(I), np.zeros(I)
for f in cfg.net.inputProcessing:
    if f.function == "l2normalize":
        continue
    assert f.function in ("mapminmax", "mapstd")
    fo = (fo - f.xOffsets.astype(np.float64)) * f.gains.astype(np.float64) + float(f.y)
    fa = fa * f.gains.astype(np.float64)
assert L0.transferFunction == "TanSig" and L1.transferFunction == "PureLin"
sc, w1s = 2.8853900817779268, -2.0
W = L0.weights.reshape(H, I).astype(np.float64)
Wq = np.zeros((H, 320))
Wq[:, :I] = bf16((sc * W * fa[None, :]).astype(np.float32)).astype(np.float64)
bias = (sc * (L0.biases.astype(np.float64) + W @ fo)).astype(np.float32).astype(np.float64)
w1q = (w1s * L1.weights.reshape(H).astype(np.float64)).astype(np.float32).astype(np.float64)
NCH = H // 32



def operands(cols, c, e0, n):
    """bf16 operands of evaluations e0 .. e0 + n - 1 of channel c, [n][320] float64, as the kernel's front makes them"""
    frames = cols[c, e0:e0 + n + T - 1, :].cpu().numpy().astype(np.float32)          # [n + T - 1][F]
    css = np.zeros(len(frames), np.float32)
    for b in range(F):                                                              # a = fmaf(x, x, a)
        css = (frames[:, b].astype(np.float64) ** 2 + css.astype(np.float64)).astype(np.float32)
    out = np.zeros((n, 320))
    for i in range(n):
        ss = np.float32(0.0)
        for tt in range(T):
            ss = np.float32(ss + css[i + tt])
        rinv = np.float32(1.0) / np.sqrt(ss, dtype=np.float32)
        v = frames[i:i + T].reshape(-1)
        out[i, :I] = bf16((v * rinv).astype(np.float32)).astype(np.float64)
    return out


def r_of(acc):
    return 1.0 / (np.exp2(acc) + 1.0)


