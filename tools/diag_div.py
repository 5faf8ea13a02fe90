import numpy as np, sys
sys.path.insert(0, '.')
from tests.test_gpu_primitives import _plane_arith, _same_bits, _division_cases
num3, den, c, k = _division_cases(0)
n = len(den)
q3, ck, _sq = _plane_arith(num3, den, c, k)
with np.errstate(all='ignore'):
    want = num3 / den[:, None]
bad = ~_same_bits(q3, want).reshape(n, 3)
print('bad', bad.sum(), 'of', bad.size)
i, j = np.nonzero(bad)
for t in range(min(25, len(i))):
    a, b = num3[i[t], j[t]], den[i[t]]
    print(f'a={a!r} b={b!r} ({np.frexp(b)[1]}) got={q3[i[t], j[t]]!r} want={want[i[t], j[t]]!r} row={num3[i[t]].tolist()}')
with np.errstate(all='ignore'):
    badc = ~_same_bits(ck, c / np.float64(k))
print('bad c', badc.sum(), 'k', k)
for t in np.nonzero(badc)[0][:10]:
    print(repr(c[t]), repr(ck[t]), repr(c[t]/k))
