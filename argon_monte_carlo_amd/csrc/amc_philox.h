// amc_philox.h — Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) for the opt-in device-side random draws: the energised
// walls' re-emission (amc_energised.hip) and the synthetic initial conditions (amc_ic.hip).  Pinned against the published
// known-answer vectors through tests/philox_ref.py.
#pragma once

// counter-based: a draw depends only on (seed, counter) — not on the order of the work items, the shard layout or the
// launch geometry
__device__ inline void philox_round(unsigned int (&c)[4], const unsigned int (&k)[2])
{
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k[0], n1 = (unsigned int)p1;
    const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k[1], n3 = (unsigned int)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ inline void philox4x32_10(unsigned int (&c)[4], unsigned long long seed)
{
    unsigned int k[2] = {(unsigned int)seed, (unsigned int)(seed >> 32)};
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
}
