"""CPU model of the report pass's fp32 filter (telescope_amd/csrc/tsem_report_pack.h, k_report_pack32): the same arithmetic in numpy
float32 — scaled score table and pi*theta words, products with the entry's position in the low mantissa bits, fp32 sums in an arbitrary
order — and the same decision rule; on rows built to sit NEAR its margins (runner-up within 2^-12 .. 2^-24 of the maximum, z_max within
2^-11 .. 2^-24 of conf_prob) every row the filter calls DECIDED must come out as the exact fp64 arithmetic says: one best hit (no other
numerator within 2^-17), and z_max on the decided side of conf_prob by more than the rounding of ANY order of additions.  What the
filter leaves undecided goes to the exact kernel on the device; here it is only counted.  No GPU, no oracle: this pins the error
analysis in the header comment."""
import numpy as np
import pytest

RP_FLOOR, RP_NEAR, RP_TMARGIN = 2.0 ** -40, 2.0 ** -16, 2.0 ** -15


def _model_row(q, c, thresh, E, rng, sq):
    """-> (decided, winner index, passes) like the kernel's last lane."""
    n = len(q)
    q32 = np.float32(np.ldexp(q, -sq))
    c32 = np.float32(np.ldexp(c, 60))
    p = (q32 * c32).astype(np.float32)                                   # fl32 of the product of two fl32
    bits = (p.view(np.uint32) & np.uint32(~np.uint32(E - 1))) | (np.arange(n) % E).astype(np.uint32)
    p = bits.view(np.float32)
    # the row sum: fp32 additions in an arbitrary association (lane-local left to right, then a scan over lanes on the device)
    order = rng.permutation(n)
    s = np.float32(0)
    for v in p[order]:
        s = np.float32(s + v)
    srt = np.sort(bits)[::-1]
    m1 = srt[0:1].view(np.float32)[0]
    m2 = srt[1:2].view(np.float32)[0] if n > 1 else np.float32(0)
    ts = np.float32(np.float32(thresh) * s)
    passes = m1 > np.float32(ts * np.float32(1.0 + RP_TMARGIN))
    fails = m1 < np.float32(ts * np.float32(1.0 - RP_TMARGIN))
    decided = (m1 >= np.float32(RP_FLOOR)) and (m2 < np.float32(m1 * np.float32(1.0 - RP_NEAR))) and (passes or fails)
    return bool(decided), int(np.argmax(bits)), bool(passes)


@pytest.mark.parametrize('E', [8, 16])
def test_decided_rows_agree_with_the_exact_arithmetic(E):
    rng = np.random.RandomState(5 + E)
    max_score = 300
    lut = np.expm1(np.arange(max_score + 1) / max_score * 100.0)
    sq = int(np.floor(np.log2(lut.max()))) - 40
    decided = undecided = 0
    for trial in range(6000):
        n = int(rng.choice([2, 3, 5, 8, 9, 16, 17, 40, 64, 130, 400]))
        code = rng.randint(139, 301, n)
        q = lut[code]
        c = np.exp(rng.uniform(-60, -2, n)) if trial % 7 else np.exp(rng.uniform(-330, -200, n))   # (some rows far down: underflow of the words)
        num = q * c
        kind = trial % 4
        i = int(np.argmax(num))
        if kind == 1 and n > 1:                                # a runner-up a hair below the maximum
            j = (i + 1) % n
            cj = num[i] * (1.0 - 2.0 ** -rng.uniform(12, 24)) / q[j]
            if cj <= 1.0:                                      # (pi * theta is at most 1)
                c[j] = cj
        num = q * c
        S = float(np.sum(num))
        zmax = float(num.max() / S) if S > 0 else 0.0
        thresh = 0.9
        if kind == 2 and 0.52 < zmax < 0.9999:                  # conf_prob a hair beside z_max
            thresh = min(0.99999, max(0.5101, zmax * (1.0 + rng.choice([-1, 1]) * 2.0 ** -rng.uniform(11, 24))))
        d, w, p = _model_row(q, c, thresh, E, rng, sq)
        if not d:
            undecided += 1
            continue
        decided += 1
        srt = np.sort(num)[::-1]
        assert w == int(np.argmax(num)), (trial, 'winner')
        assert srt[1] < srt[0] * (1.0 - 2.0 ** -17), (trial, 'runner-up inside the near-tie band of a decided row')
        # z_max against conf_prob: the reference's z_max = fl(M fl(1 / S')) with S' ANY order of fp64 additions lies within ~2^-50 of M / S
        assert (zmax > thresh * (1.0 + 2.0 ** -17)) if p else (zmax < thresh * (1.0 - 2.0 ** -17)), (trial, zmax, thresh, p)
    assert decided > 2500 and undecided > 500, (decided, undecided)      # both sides of the margins were exercised


def test_rows_far_below_the_floor_are_never_decided():
    rng = np.random.RandomState(1)
    lut = np.expm1(np.arange(301) / 300 * 100.0)
    sq = int(np.floor(np.log2(lut.max()))) - 40
    for _ in range(200):
        n = rng.randint(2, 40)
        q = lut[rng.randint(1, 60, n)]                          # small scores ...
        c = np.exp(rng.uniform(-700, -120, n))                  # ... times vanishing pi*theta: below 2^-40 in the filter's units
        d, _, _ = _model_row(q, c, 0.9, 16, rng, sq)
        assert not d
