// amc_grid.hip — candidate detection for the p-p sweep (the parallel half of Pore:520-549 / Cube:231-336).
//
// The reference finds colliding pairs by an O(n^2) loop inside each ~21 nm cell.  Every pair closer than
// collision_range shares at least one reference cell, so ANY exact close-pair detector produces a
// superset of the pairs the reference can hit; the ordered resolve (amc_resolve.hip) then re-tests each pair with
// the reference's own cell membership and loop order.  Two detectors:
//
//   binned   : counting sort of the particles into cells of edge h ~ mean spacing (>= collision_range), x fastest,
//              then each particle probes only the cells its collision_range box overlaps (1.7 on average).  O(N) work, HBM/L2-bound.  Algorithmic traffic 24 B/particle (positions read once);
//              the sorted copy (28 B written + read) and the cell tables are implementation overhead.
//   all-pairs: LDS-tiled j-block (256 particles = 6 KB) against 256 i-particles in registers, upper triangle of
//              tiles only — the kernel the reference's pairwise_particles_in_cell maps to directly; fp64-VALU-bound
//              (9 flop per pair), used for single cells (amc_pairwise_cell), small N and cross-validation.
//
// A pair is emitted when d^2 < collision_range^2 * (1 + 1e-9): a superset of the reference's
// `sqrt(d^2) < collision_range` test, which the resolve re-evaluates exactly.
#include "amc_grid_dev.h"

#define AMC_CR2_INFLATE (1.0 + 1.0e-9)

// ---- binning -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bin_count(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, long long n, amc_grid G,
                                                   int *__restrict__ cell_count, int *__restrict__ cid,
                                                   int *__restrict__ rank, amc_dev_counters *cnt)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int cx, cy, cz;
    amc_grid_coords(G, x[p], y[p], z[p], cx, cy, cz);
    bool outside = false;
    const int c = amc_grid_cell(G, cx, cy, cz, &outside);
    if (outside) atomicOr(&cnt->flags, 8ULL);
    cid[p] = c;
    rank[p] = atomicAdd(&cell_count[c], 1);
}

// exclusive scan, three small kernels: per-block scan of 4096 items -> scan of block sums -> add back
#define SCAN_T 1024
#define SCAN_ITEMS 4
#define SCAN_BLOCK (SCAN_T * SCAN_ITEMS)

__device__ inline int block_exclusive_scan_1024(int v, int *total)
{
    __shared__ int wsum[SCAN_T / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (w == 0) {
        int s = (lane < SCAN_T / 64) ? wsum[lane] : 0;
        for (int o = 1; o < SCAN_T / 64; o <<= 1) {
            int t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane < SCAN_T / 64) wsum[lane] = s;
    }
    __syncthreads();
    const int base = (w > 0) ? wsum[w - 1] : 0;
    if (total) *total = wsum[SCAN_T / 64 - 1];
    __syncthreads();
    return base + inc - v;
}

// reads the per-cell counts AND zeroes them for the next step's counting pass (saves a memset launch per step)
__global__ __launch_bounds__(SCAN_T) void k_scan_block(int *__restrict__ in, int *__restrict__ out, int n,
                                                       int *__restrict__ block_sums)
{
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        if (base + k < n) in[base + k] = 0;
        s += v[k];
    }
    int total;
    int ex = block_exclusive_scan_1024(s, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// second (last) scan kernel: every block sums the raw totals of the blocks before it (a few hundred ints) and adds
// that offset to its tile; the last block also writes the grand total behind the table
__global__ __launch_bounds__(SCAN_T) void k_scan_add(int *__restrict__ out, int n, const int *__restrict__ block_sums, int nb)
{
    __shared__ int s_off, s_tot;
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_T) part += block_sums[b];
    int total;
    block_exclusive_scan_1024(part, &total);
    if (threadIdx.x == 0) { s_off = total; s_tot = total + block_sums[blockIdx.x]; }
    __syncthreads();
    const int add = s_off;
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) out[base + k] += add;
    if ((int)blockIdx.x == nb - 1 && threadIdx.x == 0) out[n] = s_tot;
}

__global__ __launch_bounds__(256) void k_bin_scatter(const double *__restrict__ x, const double *__restrict__ y,
                                                     const double *__restrict__ z, long long n,
                                                     const int *__restrict__ cid, const int *__restrict__ rank,
                                                     const int *__restrict__ cell_start, double4 *__restrict__ sp)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int s = cell_start[cid[p]] + rank[p];
    sp[s] = make_double4(x[p], y[p], z[p], amc_sp_pack((int)p));      // one 32-byte scattered store per particle
}

// ---- binned detection: one thread per sorted particle, half stencil ------------------------------------------------
// A candidate is stored as (i, j) with i > j plus a copy of both particles' state in a SoA table (cst[e][k], e = 0..10
// particle j, 11..21 particle i): the single-workgroup resolve kernel then reads coalesced rows instead of issuing
// 22 scattered loads per pair from one CU.  Returns the candidate's slot (or -1 on overflow).
AMC_DEV int amc_push_candidate(int a, int b, int *cand_i, int *cand_j, int max_cand, amc_dev_counters *cnt)
{
    const unsigned int k = atomicAdd(&cnt->cand_count, 1u);
    if (k < (unsigned)max_cand) {
        cand_i[k] = a > b ? a : b;
        cand_j[k] = a > b ? b : a;
        return (int)k;
    }
    atomicOr(&cnt->flags, 1ULL);
    return -1;
}

// state gather for the candidates found by the lanes of this wave, done by the WHOLE wave: lane e < 22 moves element
// e of the pair (11 per particle), so a candidate costs one load + one store instruction instead of 44 serial ones
AMC_DEV void amc_wave_gather(unsigned long long found, int my_k, int my_i, int my_j, int max_cand,
                             const amc_state &S, double *cst)
{
    const int lane = threadIdx.x & 63;
    while (found) {
        const int src = __ffsll((long long)found) - 1;
        found &= found - 1;
        const int k = __shfl(my_k, src, 64);
        const int pi = __shfl(my_i, src, 64), pj = __shfl(my_j, src, 64);     // (i > j), straight from the finder's registers
        if (k < 0 || lane >= 22) continue;
        const int w = lane / 11, e = lane % 11;
        const int p = w ? pi : pj;
        double v;
        switch (e) {
        case 0: v = S.x[p]; break; case 1: v = S.y[p]; break; case 2: v = S.z[p]; break;
        case 3: v = S.vx[p]; break; case 4: v = S.vy[p]; break; case 5: v = S.vz[p]; break;
        case 6: v = S.d[p]; break; case 7: v = S.dx[p]; break; case 8: v = S.dy[p]; break; case 9: v = S.dz[p]; break;
        default: v = S.flag[p] ? 1.0 : 0.0; break;
        }
        cst[(size_t)lane * (size_t)max_cand + k] = v;
    }
}

__global__ __launch_bounds__(256) void k_detect_binned(amc_grid G, const double4 *__restrict__ sp,
                                                       const int *__restrict__ cell_start, long long n, double cr2i,
                                                       double cr_probe, int *cand_i, int *cand_j, int max_cand,
                                                       amc_dev_counters *cnt, amc_state S, double *cst)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int my_k = -1, my_i = 0, my_j = 0;   // a lane finds at most a handful of pairs; the (rare) 2nd+ ones are gathered right away
    if (s < n) {
        const double4 me = sp[s];
        const int me_idx = amc_sp_index(me);
        // cells overlapped by my collision_range box (1.7 on average, 8 at most); each pair is seen from both ends
        // and emitted by the one that comes later in the sorted order
        int c_lo[4], c_hi[4];
        const int nc = amc_grid_box_ranges(G, me.x, me.y, me.z, cr_probe, c_lo, c_hi);
        for (int k = 0; k < nc; k++) {
            const int q0 = cell_start[c_lo[k]], q1 = cell_start[c_hi[k] + 1];
            for (int q = q0; q < q1; q++) {
                if (q <= (int)s) continue;
                const double4 o = sp[q];
                const double ex = o.x - me.x, ey = o.y - me.y, ez = o.z - me.z;
                if (ex * ex + ey * ey + ez * ez < cr2i) {
                    const int kk = amc_push_candidate(me_idx, amc_sp_index(o), cand_i, cand_j, max_cand, cnt);
                    if (my_k >= 0 && kk >= 0) {
                        // second pair of this lane: gather it alone (divergent, rare)
                        const int oi2 = amc_sp_index(o);
                        const int hi2 = me_idx > oi2 ? me_idx : oi2, lo2 = me_idx > oi2 ? oi2 : me_idx;
                        for (int lane = 0; lane < 22; lane++) {
                            const int p = lane / 11 ? hi2 : lo2;
                            const int e = lane % 11;
                            const double v = e == 0 ? S.x[p] : e == 1 ? S.y[p] : e == 2 ? S.z[p] : e == 3 ? S.vx[p] :
                                             e == 4 ? S.vy[p] : e == 5 ? S.vz[p] : e == 6 ? S.d[p] : e == 7 ? S.dx[p] :
                                             e == 8 ? S.dy[p] : e == 9 ? S.dz[p] : (S.flag[p] ? 1.0 : 0.0);
                            cst[(size_t)lane * (size_t)max_cand + kk] = v;
                        }
                    } else {
                        my_k = kk;
                        const int oi = amc_sp_index(o);
                        my_i = me_idx > oi ? me_idx : oi;
                        my_j = me_idx > oi ? oi : me_idx;
                    }
                }
            }
        }
    }
    const unsigned long long found = __ballot(my_k >= 0);
    if (found) amc_wave_gather(found, my_k, my_i, my_j, max_cand, S, cst);
}

// ---- all-pairs detection: LDS tile of 256 j-particles against 256 i-particles in registers ---------------------------
#define AP_T 256
__global__ __launch_bounds__(AP_T) void k_detect_allpairs(const double *__restrict__ x, const double *__restrict__ y,
                                                          const double *__restrict__ z, int n, int ntiles, double cr2i,
                                                          int *cand_i, int *cand_j, int max_cand,
                                                          amc_dev_counters *cnt, amc_state S, double *cst)
{
    // blockIdx.x enumerates the lower triangle of tile pairs: (bi, bj) with bj <= bi
    int bi = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long long)(bi + 1) * (bi + 2) / 2 <= (long long)blockIdx.x) bi++;
    while ((long long)bi * (bi + 1) / 2 > (long long)blockIdx.x) bi--;
    const int bj = (int)(blockIdx.x - (long long)bi * (bi + 1) / 2);
    if (bi >= ntiles) return;
    __shared__ double tx[AP_T], ty[AP_T], tz[AP_T];
    const int j0 = bj * AP_T;
    const int jj = j0 + threadIdx.x;
    tx[threadIdx.x] = (jj < n) ? x[jj] : 0.0;
    ty[threadIdx.x] = (jj < n) ? y[jj] : 0.0;
    tz[threadIdx.x] = (jj < n) ? z[jj] : 0.0;
    __syncthreads();
    const int i = bi * AP_T + threadIdx.x;
    if (i >= n) return;
    const double xi = x[i], yi = y[i], zi = z[i];
    int jmax = n - j0;
    if (jmax > AP_T) jmax = AP_T;
    if (bi == bj && jmax > (int)threadIdx.x) jmax = threadIdx.x;   // j < i inside the diagonal tile
    for (int k = 0; k < jmax; k++) {
        const double ex = tx[k] - xi, ey = ty[k] - yi, ez = tz[k] - zi;
        const double d2 = ex * ex + ey * ey + ez * ez;
        if (d2 < cr2i) {
            const int kk = amc_push_candidate(i, j0 + k, cand_i, cand_j, max_cand, cnt);
            if (kk >= 0)
                for (int lane = 0; lane < 22; lane++) {
                    const int p = lane / 11 ? i : (j0 + k);
                    const int e = lane % 11;
                    const double v = e == 0 ? S.x[p] : e == 1 ? S.y[p] : e == 2 ? S.z[p] : e == 3 ? S.vx[p] : e == 4 ? S.vy[p] :
                                     e == 5 ? S.vz[p] : e == 6 ? S.d[p] : e == 7 ? S.dx[p] : e == 8 ? S.dy[p] : e == 9 ? S.dz[p] :
                                     (S.flag[p] ? 1.0 : 0.0);
                    cst[(size_t)lane * (size_t)max_cand + kk] = v;
                }
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------
hipError_t amc_launch_bin_clear(amc_ctx *c)
{
    // the counters are zeroed once at creation and re-zeroed by k_scan_block every time they are consumed
    (void)c;
    return hipSuccess;
}

hipError_t amc_launch_bin(amc_ctx *c, bool counted)
{
    if (c->allpairs) return hipSuccess;
    const long long n = c->n;
    const int nc = c->G.ncells;
    hipError_t e;
    if (!counted) {
        amc_prof_begin(c, AMC_K_BIN_COUNT);
        e = amc_launch_bin_clear(c);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_bin_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->S.x, c->S.y, c->S.z, n,
                           c->G, c->B.cell_count, c->B.cid, c->B.rank, c->d_cnt);
        amc_prof_end(c);
    }
    const int nb = (nc + SCAN_BLOCK - 1) / SCAN_BLOCK;
    amc_prof_begin(c, AMC_K_BIN_SCAN);
    hipLaunchKernelGGL(k_scan_block, dim3(nb), dim3(SCAN_T), 0, c->stream, c->B.cell_count, c->B.cell_start, nc,
                       c->scan_tmp);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_T), 0, c->stream, c->B.cell_start, nc, c->scan_tmp, nb);
    amc_prof_end(c);
    amc_prof_begin(c, AMC_K_BIN_SCATTER);
    hipLaunchKernelGGL(k_bin_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->S.x, c->S.y, c->S.z,
                       n, c->B.cid, c->B.rank, c->B.cell_start, c->B.sp);
    amc_prof_end(c);
    return hipGetLastError();
}

hipError_t amc_launch_detect(amc_ctx *c)
{
    const long long n = c->n;
    const double cr2i = c->P.collision_range * c->P.collision_range * AMC_CR2_INFLATE;
    amc_prof_begin(c, AMC_K_DETECT);
    if (c->allpairs) {
        const int ntiles = (int)((n + AP_T - 1) / AP_T);
        const long long nblocks = (long long)ntiles * (ntiles + 1) / 2;
        if (nblocks > 0)
            hipLaunchKernelGGL(k_detect_allpairs, dim3((unsigned)nblocks), dim3(AP_T), 0, c->stream, c->S.x, c->S.y,
                               c->S.z, (int)n, ntiles, cr2i, c->W.cand_i, c->W.cand_j, c->W.max_cand, c->d_cnt, c->S, c->W.cst);
    } else {
        hipLaunchKernelGGL(k_detect_binned, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->G, c->B.sp, c->B.cell_start, n, cr2i, c->P.collision_range * 1.000001, c->W.cand_i, c->W.cand_j,
                           c->W.max_cand, c->d_cnt, c->S, c->W.cst);
    }
    amc_prof_end(c);
    return hipGetLastError();
}
