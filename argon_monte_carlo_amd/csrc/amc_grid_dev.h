// amc_grid_dev.h — device helpers of the detection grid (uniform cells, per-z-layer square window).
#pragma once
#include "amc_internal.h"

AMC_DEV int amc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

AMC_DEV void amc_grid_coords(const amc_grid &G, double x, double y, double z, int &cx, int &cy, int &cz)
{
    cx = amc_clampi((int)floor((x - G.x0) / G.h), 0, G.gx - 1);
    cy = amc_clampi((int)floor((y - G.y0) / G.h), 0, G.gy - 1);
    cz = amc_clampi((int)floor((z - G.z0) / G.h), 0, G.gz - 1);
}

// linear id of the cell that STORES a particle with coordinates (cx,cy,cz): coordinates outside the layer's window
// are clamped into it (cannot happen for in-bounds particles; *outside is set so the caller can flag it)
AMC_DEV int amc_grid_cell(const amc_grid &G, int cx, int cy, int cz, bool *outside)
{
    const int lo = G.lay_lo[cz], n = G.lay_n[cz];
    int lx = cx - lo, ly = cy - lo;
    if (lx < 0 || lx >= n || ly < 0 || ly >= n) {
        if (outside) *outside = true;
        lx = amc_clampi(lx, 0, n - 1);
        ly = amc_clampi(ly, 0, n - 1);
    }
    return G.lay_off[cz] + ly * n + lx;
}

// cells (cx-1..cx+1, cy, cz) clipped to the layer's window: contiguous ids [c_lo, c_hi]; false if the row is empty
AMC_DEV bool amc_grid_row(const amc_grid &G, int cx, int cy, int cz, int &c_lo, int &c_hi)
{
    if (cz < 0 || cz >= G.gz) return false;
    const int lo = G.lay_lo[cz], n = G.lay_n[cz];
    const int ly = cy - lo;
    if (ly < 0 || ly >= n) return false;
    // a stored particle may have been clamped into the window edge; querying from the true coordinates keeps
    // in-window pairs exact
    int x_lo = cx - 1 - lo, x_hi = cx + 1 - lo;
    if (x_lo < 0) x_lo = 0;
    if (x_hi > n - 1) x_hi = n - 1;
    if (x_lo > x_hi) return false;
    const int base = G.lay_off[cz] + ly * n;
    c_lo = base + x_lo;
    c_hi = base + x_hi;
    return true;
}

// The stored cells that can hold a particle within distance r of (x,y,z): the cells overlapped by the box
// [x-r,x+r] x [y-r,y+r] x [z-r,z+r] — at most 2 per axis because h >= r — after the same monotone clamps that
// amc_grid_coords / amc_grid_cell apply to particles.  Writes up to 8 distinct cell ids, returns their number.
AMC_DEV int amc_grid_box_cells(const amc_grid &G, double x, double y, double z, double r, int *cells)
{
    int x0, y0, z0, x1, y1, z1;
    amc_grid_coords(G, x - r, y - r, z - r, x0, y0, z0);
    amc_grid_coords(G, x + r, y + r, z + r, x1, y1, z1);
    int n = 0;
    for (int cz = z0; cz <= z1; cz++)
        for (int cy = y0; cy <= y1; cy++)
            for (int cx = x0; cx <= x1; cx++) {
                const int c = amc_grid_cell(G, cx, cy, cz, nullptr);
                bool dup = false;
                for (int k = 0; k < n; k++) dup |= (cells[k] == c);
                if (!dup && n < 8) cells[n++] = c;
            }
    return n;
}
