"""Plain-Python Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) —
test infrastructure: pinned to the published known-answer vectors in test_host.py, then used to check what the device
generator of the opt-in energised-wall mode produced."""
M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    c = [int(v) & MASK for v in counter]
    k = [int(v) & MASK for v in key]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k[1]) & MASK, p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return c


def direction_draw(seed, particle, step, case, attempt):
    """(costheta, phi, sign) exactly as k_temp_sample forms them from one Philox block."""
    import math
    c = philox4x32_10([particle, step, (case << 16) | attempt, 0x414d4331], [seed & MASK, (seed >> 32) & MASK])
    u1 = (((c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0)
    u2 = (((c[2] << 32) | c[3]) >> 12) * (1.0 / 4503599627370496.0)
    return -1.0 + 2.0 * u1, math.pi * u2, (1.0 if (c[3] & 1) else -1.0)


def ic_uniforms(seed, particle):
    """The eight uniform doubles k_ic (amc_ic.hip) forms for one particle: Philox blocks 0..3 with counter
    (particle, block, 0, "AMCI"), two 53-bit numbers per block."""
    out = []
    for blk in range(4):
        c = philox4x32_10([particle, blk, 0, 0x414d4349], [seed & MASK, (seed >> 32) & MASK])
        out.append((((c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0))
        out.append((((c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0))
    return out
