"""CPU emulation of the all-pairs kernel's accumulation orders on the alpha_full-shaped fixture (round 3, DESIGN.md
section 2): per 32-deep stage three fp16-product sums (hi x lo, lo x hi, hi x hi), each added to an fp32 accumulator with
ONE rounding (the hardware rounds more often: measured errors are ~3x these) -
  single : one fp32 accumulator for the whole dot product
  two    : the shipped two-level sum (flush every 16 stages)
  norm   : chunk half-norms subtracted on the VALU every 16 stages
  slot   : norm GROUPS in the operands (the MFMA subtracts them), chunks of 16 / 32 / 64 / 128 stages
Prints max / rms of d - d_truth over 400 near pairs (same patch of two images) and 100 far pairs."""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tests')
from _synth import clustered_embeddings
Z, _ = clustered_embeddings(3, 784, 4096, 3, salt=41)
def split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)
rng = np.random.default_rng(0)
A = Z[0]; B = Z[1]
idx = rng.choice(784, 400, replace=False)
pairs = [(p, p) for p in idx] + [(p, (p * 7 + 3) % 784) for p in idx[:100]]
def run(a, b, scheme):
    ah, al = split(a); bh, bl = split(b)
    a64 = ah + al; b64 = bh + bl           # what the kernel sees
    truth_d2 = ((a64 - b64) ** 2).sum()
    na = (a64 * a64); nb = (b64 * b64)
    acc = np.float32(0); tot = np.float32(0)
    for s in range(128):
        sl = slice(32 * s, 32 * s + 32)
        for (x, y) in ((ah, bl), (al, bh), (ah, bh)):
            acc = np.float32(np.float64(acc) + (x[sl] * y[sl]).sum())
        if s % 16 == 15:
            if scheme == 'two':
                tot = np.float32(tot + acc); acc = np.float32(0)
            elif scheme == 'norm':
                c = slice(512 * (s // 16), 512 * (s // 16) + 512)
                half = np.float32(np.float32(na[c].sum()) * np.float32(0.5)) + np.float32(np.float32(nb[c].sum()) * np.float32(0.5))
                acc = np.float32(acc - np.float32(half))
    if scheme == 'norm':
        d2 = np.float32(-2.0) * acc
    else:
        dot = np.float32(tot + acc) if scheme == 'two' else acc
        ksq = np.float32(na.sum()); qs = np.float32(nb.sum())
        d2 = np.float32(np.float32(ksq - dot) + np.float32(qs - dot))
    d2 = max(float(d2), 0.0)
    return np.sqrt(d2) - np.sqrt(truth_d2), np.sqrt(truth_d2)
for scheme in ('single', 'two', 'norm'):
    e = np.array([run(A[p], B[q], scheme)[0] for (p, q) in pairs])
    nn, far = e[:400], e[400:]
    print(scheme, 'NN pairs: max %.2e rms %.2e mean %.2e | far pairs: max %.2e rms %.2e' % (np.abs(nn).max(), np.sqrt((nn**2).mean()), nn.mean(), np.abs(far).max(), np.sqrt((far**2).mean())))
print('d NN ~', run(A[5], B[5], 'two')[1], ' far ~', run(A[5], B[38], 'two')[1])

def run2(a, b, stages_per_chunk, pre):
    ah, al = split(a); bh, bl = split(b)
    a64 = ah + al; b64 = bh + bl
    truth_d2 = ((a64 - b64) ** 2).sum()
    na = (a64 * a64); nb = (b64 * b64)
    acc = np.float32(0)
    W = 32 * stages_per_chunk
    def half(ci):
        c = slice(W * ci, W * ci + W)
        return np.float32(np.float32(np.float32(na[c].sum()) * np.float32(0.5)) + np.float32(np.float32(nb[c].sum()) * np.float32(0.5)))
    for s in range(128):
        if pre and s % stages_per_chunk == 0:
            acc = np.float32(acc - half(s // stages_per_chunk))
        sl = slice(32 * s, 32 * s + 32)
        for (x, y) in ((ah, bl), (al, bh), (ah, bh)):
            acc = np.float32(np.float64(acc) + (x[sl] * y[sl]).sum())
        if (not pre) and s % stages_per_chunk == stages_per_chunk - 1:
            acc = np.float32(acc - half(s // stages_per_chunk))
    d2 = max(float(np.float32(-2.0) * acc), 0.0)
    return np.sqrt(d2) - np.sqrt(truth_d2)
for spc, pre in ((16, False), (16, True), (8, False), (8, True), (4, True)):
    e = np.array([run2(A[p], B[q], spc, pre) for (p, q) in pairs])
    nn, far = e[:400], e[400:]
    print('norm chunk=%d stages pre=%s' % (spc, pre), 'NN: max %.2e rms %.2e | far: max %.2e rms %.2e' % (np.abs(nn).max(), np.sqrt((nn**2).mean()), np.abs(far).max(), np.sqrt((far**2).mean())))

e_two = np.array([run(A[p], B[p], 'two')[0] for p in range(0, 784)])
e_nrm = np.array([run2(A[p], B[p], 16, False) for p in range(0, 784)])
na_all = (A.astype(np.float64)**2).sum(1)
for name, e in (('two', e_two), ('norm16', e_nrm)):
    q = np.quantile(np.abs(e), [0.5, 0.9, 0.99, 1.0])
    worst = np.argsort(-np.abs(e))[:5]
    print(name, 'quantiles', q, 'worst patches', worst, 'their |a|^2', na_all[worst].round(0), 'errs', e[worst])

print('---- norm-slot scheme (MFMA subtracts the chunk norms: pass1 -nb/2, pass2 -na/2, pass3 +3)')
def run3(a, b, spc):
    ah, al = split(a); bh, bl = split(b)
    a64 = ah + al; b64 = bh + bl
    truth_d2 = ((a64 - b64) ** 2).sum()
    na = (a64 * a64); nb = (b64 * b64)
    acc = np.float32(0)
    W = 32 * spc
    nchunks = (128 + spc - 1) // spc
    def parts(x):   # exact 3-term fp16 expansion of fp32(x)
        x = np.float64(np.float32(x)); p1 = np.float64(np.float16(x)); r = x - p1; p2 = np.float64(np.float16(r)); p3 = np.float64(np.float16(r - p2)); return p1 + p2 + p3
    for s in range(128):
        sl = slice(32 * s, 32 * s + 32)
        for (x, y) in ((ah, bl), (al, bh), (ah, bh)):
            acc = np.float32(np.float64(acc) + (x[sl] * y[sl]).sum())
        if s % spc == spc - 1 or s == 127:
            ci = s // spc; c = slice(W * ci, W * ci + W)
            acc = np.float32(np.float64(acc) + parts(-0.5 * nb[c].sum()))
            acc = np.float32(np.float64(acc) + parts(-0.5 * na[c].sum()))
            acc = np.float32(np.float64(acc) + 3.0)
    d2 = max(float(np.float32(-2.0) * np.float32(acc - np.float32(3.0 * nchunks))), 0.0)
    return np.sqrt(d2) - np.sqrt(truth_d2)
for spc in (16, 32, 64, 128):
    e = np.array([run3(A[p], B[q], spc) for (p, q) in pairs])
    nn, far = e[:400], e[400:]
    print('slot chunk=%d stages' % spc, 'NN: max %.2e rms %.2e | far: max %.2e rms %.2e' % (np.abs(nn).max(), np.sqrt((nn**2).mean()), np.abs(far).max(), np.sqrt((far**2).mean())))
