// amc_grid_dev.h — device helpers of the detection grid (uniform cells, per-z-layer square window, per-cell lists).
#pragma once
#include "amc_internal.h"

AMC_DEV int amc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

AMC_DEV void amc_grid_coords(const amc_grid &G, double x, double y, double z, int &cx, int &cy, int &cz)
{
    cx = amc_clampi((int)floor((x - G.x0) * G.inv_h), 0, G.gx - 1);
    cy = amc_clampi((int)floor((y - G.y0) * G.inv_h), 0, G.gy - 1);
    cz = amc_clampi((int)floor((z - G.z0) * G.inv_h), 0, G.gz - 1);
}

// linear id of the cell that STORES a particle with coordinates (cx,cy,cz): coordinates outside the layer's window
// are clamped into it (cannot happen for in-bounds particles; *outside is set so the caller can flag it)
AMC_DEV int amc_grid_cell(const amc_grid &G, int cx, int cy, int cz, bool *outside)
{
    if (G.uniform) return (cz * G.gy + cy) * G.gx + cx;      // coordinates are already clamped into the full window
    const int lo = G.lay_lo[cz], n = G.lay_n[cz];
    int lx = cx - lo, ly = cy - lo;
    if (lx < 0 || lx >= n || ly < 0 || ly >= n) {
        if (outside) *outside = true;
        lx = amc_clampi(lx, 0, n - 1);
        ly = amc_clampi(ly, 0, n - 1);
    }
    return G.lay_off[cz] + ly * n + lx;
}

// The stored cells that can hold a particle within distance r (<= h/2) of (x,y,z), as up to four runs of x-adjacent
// cells [c_lo, c_hi] (adjacent entries of the head table).  One floor per axis: with f = fractional position inside the
// cell, the lower neighbour is needed iff f < r/h and the upper one iff f > 1 - r/h; the thresholds are inflated by
// 1e-6 cells, far above the rounding of (v - v0) * inv_h (~1e-13 cells), so the set is a superset of the exact one
// under the same monotone clamps that binning applies.  Returns the number of runs.
AMC_DEV int amc_grid_box_ranges(const amc_grid &G, double x, double y, double z, double r, int *c_lo, int *c_hi)
{
    const double ux = (x - G.x0) * G.inv_h, uy = (y - G.y0) * G.inv_h, uz = (z - G.z0) * G.inv_h;
    const double fx0 = floor(ux), fy0 = floor(uy), fz0 = floor(uz);
    const double t = r * G.inv_h + 1.0e-6;
    const int cx = amc_clampi((int)fx0, 0, G.gx - 1), cy = amc_clampi((int)fy0, 0, G.gy - 1), cz = amc_clampi((int)fz0, 0, G.gz - 1);
    const double fx = ux - fx0, fy = uy - fy0, fz = uz - fz0;
    // coordinates outside the grid are clamped by binning; there the fractional test is meaningless -> take both sides
    const bool ox = (fx0 < 0) || (fx0 > G.gx - 1), oy = (fy0 < 0) || (fy0 > G.gy - 1), oz = (fz0 < 0) || (fz0 > G.gz - 1);
    const int x_lo = amc_clampi(cx - ((fx < t || ox) ? 1 : 0), 0, G.gx - 1), x_hi = amc_clampi(cx + ((fx > 1.0 - t || ox) ? 1 : 0), 0, G.gx - 1);
    const int y_lo = amc_clampi(cy - ((fy < t || oy) ? 1 : 0), 0, G.gy - 1), y_hi = amc_clampi(cy + ((fy > 1.0 - t || oy) ? 1 : 0), 0, G.gy - 1);
    const int z_lo = amc_clampi(cz - ((fz < t || oz) ? 1 : 0), 0, G.gz - 1), z_hi = amc_clampi(cz + ((fz > 1.0 - t || oz) ? 1 : 0), 0, G.gz - 1);
    int n = 0;
    for (int kz = z_lo; kz <= z_hi; kz++) {
        int lo = 0, nn = G.gx, off;
        if (G.uniform) off = kz * G.gy * G.gx;
        else { lo = G.lay_lo[kz]; nn = G.lay_n[kz]; off = G.lay_off[kz]; }
        // window clamp of this layer (binning clamps out-of-window particles into the edge cells)
        const int a = amc_clampi(x_lo - lo, 0, nn - 1), b = amc_clampi(x_hi - lo, 0, nn - 1);
        const int ya = amc_clampi(y_lo - lo, 0, nn - 1), yb = amc_clampi(y_hi - lo, 0, nn - 1);
        for (int ky = ya; ky <= yb; ky++) {
            if (n < 4) { c_lo[n] = off + ky * nn + a; c_hi[n] = off + ky * nn + b; n++; }
        }
    }
    return n;
}

// Cell number c (0..7) of the same enumeration: run c >> 1 (z-major, then y), end c & 1 (lower / upper x cell of the run);
// -1 if there is none.  For probes that spread the cells of one box over several lanes.
AMC_DEV int amc_grid_box_cell(const amc_grid &G, double x, double y, double z, double r, int c)
{
    const double ux = (x - G.x0) * G.inv_h, uy = (y - G.y0) * G.inv_h, uz = (z - G.z0) * G.inv_h;
    const double fx0 = floor(ux), fy0 = floor(uy), fz0 = floor(uz);
    const double t = r * G.inv_h + 1.0e-6;
    const int cx = amc_clampi((int)fx0, 0, G.gx - 1), cy = amc_clampi((int)fy0, 0, G.gy - 1), cz = amc_clampi((int)fz0, 0, G.gz - 1);
    const double fx = ux - fx0, fy = uy - fy0, fz = uz - fz0;
    const bool ox = (fx0 < 0) || (fx0 > G.gx - 1), oy = (fy0 < 0) || (fy0 > G.gy - 1), oz = (fz0 < 0) || (fz0 > G.gz - 1);
    const int x_lo = amc_clampi(cx - ((fx < t || ox) ? 1 : 0), 0, G.gx - 1), x_hi = amc_clampi(cx + ((fx > 1.0 - t || ox) ? 1 : 0), 0, G.gx - 1);
    const int y_lo = amc_clampi(cy - ((fy < t || oy) ? 1 : 0), 0, G.gy - 1), y_hi = amc_clampi(cy + ((fy > 1.0 - t || oy) ? 1 : 0), 0, G.gy - 1);
    const int z_lo = amc_clampi(cz - ((fz < t || oz) ? 1 : 0), 0, G.gz - 1), z_hi = amc_clampi(cz + ((fz > 1.0 - t || oz) ? 1 : 0), 0, G.gz - 1);
    int run = c >> 1;
    for (int kz = z_lo; kz <= z_hi; kz++) {
        int lo = 0, nn = G.gx, off;
        if (G.uniform) off = kz * G.gy * G.gx;
        else { lo = G.lay_lo[kz]; nn = G.lay_n[kz]; off = G.lay_off[kz]; }
        const int a = amc_clampi(x_lo - lo, 0, nn - 1), b = amc_clampi(x_hi - lo, 0, nn - 1);
        const int ya = amc_clampi(y_lo - lo, 0, nn - 1), yb = amc_clampi(y_hi - lo, 0, nn - 1);
        const int ny = yb - ya + 1;
        if (run < ny) {
            const int row = off + (ya + run) * nn;
            if (c & 1) return b != a ? row + b : -1;
            return row + a;
        }
        run -= ny;
    }
    return -1;
}

// ---- per-cell lists -----------------------------------------------------------------------------------------------------
AMC_DEV int amc_rec_next(const amc_rec &r) { return r.next; }
// the particle a list node stands for (amc_lists: nodes below n are the particles themselves)
AMC_DEV int amc_node_particle(const amc_lists &B, int node) { return node < B.n ? node : B.extra[node - B.n]; }
// position of a list record (absolute, double): what the particle was filed under, within the rounding of a float
AMC_DEV void amc_rec_pos(const amc_grid &G, const amc_rec &r, double &x, double &y, double &z)
{
    x = G.x0 + (double)r.x; y = G.y0 + (double)r.y; z = G.z0 + (double)r.z;
}
// first particle of cell c in the current epoch, or -1
AMC_DEV int amc_list_head(const amc_lists &B, int c)
{
    const unsigned long long h = B.head[c];
    return ((unsigned int)(h >> 32) == B.epoch) ? (int)(unsigned int)(h & 0xffffffffULL) : -1;
}
// push particle p (position x,y,z) on the list of its cell; writes its record.  The cell is the one of the ROUNDED position
// (so that whoever reads the record back derives the same cell); probes cover the rounding through G.cr_probe.
AMC_DEV void amc_list_insert(const amc_grid &G, const amc_lists &B, int p, double x, double y, double z, bool *outside)
{
    amc_rec r;
    r.x = (float)(x - G.x0); r.y = (float)(y - G.y0); r.z = (float)(z - G.z0);
    double rx, ry, rz;
    amc_rec_pos(G, r, rx, ry, rz);
    int cx, cy, cz;
    amc_grid_coords(G, rx, ry, rz, cx, cy, cz);
    const int c = amc_grid_cell(G, cx, cy, cz, outside);
    const unsigned long long mine = ((unsigned long long)B.epoch << 32) | (unsigned int)p;
    const unsigned long long old = atomicExch(&B.head[c], mine);
    r.next = ((unsigned int)(old >> 32) == B.epoch) ? (int)(unsigned int)(old & 0xffffffffULL) : -1;
    B.rec[p] = r;
}

// Kept lists (amc_lists): file particle p at (x, y, z).  full: every particle under its own node, as amc_list_insert, and
// cell / node remembered.  Otherwise: same cell as last step -> only the node's position is refreshed; another cell -> the
// old node is poisoned (stays linked) and a new node is taken from the WAVE's pool (count0 = what the wave had handed out
// before this step; like c_old / node — the particle's cell_of / node_of — read by the caller together with the state) and pushed on the new cell's list.  Must be called by all
// active lanes of a wave together; returns the wave's new count (the caller's first active lane stores it).
AMC_DEV int amc_list_keep(const amc_grid &G, const amc_lists &B, int p, double x, double y, double z, bool full, int wave_id,
                          int count0, int c_old, int node, bool *outside, bool *overflow)
{
    amc_rec r;
    r.x = (float)(x - G.x0); r.y = (float)(y - G.y0); r.z = (float)(z - G.z0);
    double rx, ry, rz;
    amc_rec_pos(G, r, rx, ry, rz);
    int cx, cy, cz;
    amc_grid_coords(G, rx, ry, rz, cx, cy, cz);
    const int c = amc_grid_cell(G, cx, cy, cz, outside);
    if (full) {
        const unsigned long long mine = ((unsigned long long)B.epoch << 32) | (unsigned int)p;
        const unsigned long long old = atomicExch(&B.head[c], mine);
        r.next = ((unsigned int)(old >> 32) == B.epoch) ? (int)(unsigned int)(old & 0xffffffffULL) : -1;
        B.rec[p] = r;
        B.cell_of[p] = c; B.node_of[p] = p;
        return 0;
    }
    const bool mover = c != c_old;      // (c_old, node: where the particle is filed — read by the caller together with the state)
    const unsigned long long mv = __ballot(mover);
    if (!mover) {
        typedef float v3f_ __attribute__((ext_vector_type(3)));
        v3f_ v_; v_.x = r.x; v_.y = r.y; v_.z = r.z;
        *(v3f_ *)&B.rec[node] = v_;             // (one 12-byte store; the link behind it is untouched)
        return count0 + __popcll(mv);
    }
    const int lane = (int)__lane_id();
    const int idx = count0 + __popcll(mv & ((1ULL << lane) - 1ULL));
    ((float *)&B.rec[node])[0] = __int_as_float(0x7fc00000);        // the node it leaves: NaN, no test passes; its link stays
    if (idx >= B.wave_cap) { *overflow = true; return count0 + __popcll(mv); }     // (cannot happen: see amc_lists)
    const int e = wave_id * B.wave_cap + idx, nn = B.n + e;
    B.extra[e] = p; B.node_of[p] = nn; B.cell_of[p] = c;
    const unsigned long long mine = ((unsigned long long)B.epoch << 32) | (unsigned int)nn;
    const unsigned long long old = atomicExch(&B.head[c], mine);
    r.next = ((unsigned int)(old >> 32) == B.epoch) ? (int)(unsigned int)(old & 0xffffffffULL) : -1;
    B.rec[nn] = r;
    return count0 + __popcll(mv);
}
