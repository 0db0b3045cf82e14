/*
 * amc_oracle_impl.h — body of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see amc_oracle.c header).
 *
 * Included twice by amc_oracle.c:
 *   ORC(name) = orc_pow_name, SQ(x) = pow(x, 2.0)  -> bit-for-bit the reference (NumPy *scalar* `x**2` is libm pow)
 *   ORC(name) = orc_mul_name, SQ(x) = x*x          -> the same algorithm with exact squares (what the HIP kernels use)
 * SQ is used ONLY where the reference squares a NumPy scalar; where it squares an ndarray (`x_vals**2`,
 * np.square) NumPy computes the exact product and this file writes x*x in both variants.
 * np.dot of 2-/3-vectors goes through OpenBLAS ddot whose scalar tail is an FMA chain (verified bitwise on
 * 1e5 random vectors in the build container): dot3 = fma(a2,b2, fma(a1,b1, a0*b0)).
 */

/* ------------------------------------------------------------------------------------------------------- */
/* a1 — pairwise_particles_in_cell, Pore:160-255 == Temp:215-309 == Cube:253-324                              */
/* gidx: global particle index of each member (for the path record keys) or NULL.                            */
/* amc_params.reserved1 bit0 = "count and continue" (Temp:340-342 semantics applied to every geometry — the library's
 * option for production runs; the reference's Pore/Cube scripts abort instead): events are tallied here per timestep. */
static int64_t ORC(fp_tolerated) = 0;

int ORC(pair_cell)(const amc_params *P, int64_t n, double *cont, double *contx, double *conty, double *contz,
                   uint8_t *flag, double *xs, double *ys, double *zs, double *vxs, double *vys, double *vzs,
                   const int32_t *gidx, orc_sink *sink, int32_t step, int32_t phase, int64_t cell,
                   int64_t *ncoll_out)
{
    const double cr = P->collision_range;
    const double m = P->argon_mass;
    int64_t ncoll = 0;
    int rc = 0;
    for (int64_t i = 0; i < n; i++) {          /* Pore:168 */
        for (int64_t j = 0; j < i; j++) {      /* Pore:169 */
            double x1 = xs[j], x2 = xs[i], y1 = ys[j], y2 = ys[i], z1 = zs[j], z2 = zs[i];  /* Pore:172 */
            double sep = sqrt(SQ(x2 - x1) + SQ(y2 - y1) + SQ(z2 - z1));                     /* Pore:173 */
            if (!(sep < cr)) continue;                                                       /* Pore:174 */
            double vx1 = vxs[j], vx2 = vxs[i], vy1 = vys[j], vy2 = vys[i], vz1 = vzs[j], vz2 = vzs[i]; /* 179 */
            double a = SQ(-vx2 + vx1) + SQ(-vy2 + vy1) + SQ(-vz2 + vz1);                    /* Pore:182 */
            double b = 2 * ((x2 - x1) * (-vx2 + vx1) + (y2 - y1) * (-vy2 + vy1) + (z2 - z1) * (-vz2 + vz1)); /* 183 */
            double c = SQ(x2 - x1) + SQ(y2 - y1) + SQ(z2 - z1) - SQ(cr);                    /* Pore:184 */
            double disc2 = SQ(b) - 4 * a * c;
            if (a == 0.0 || disc2 < 0.0 || a != a || disc2 != disc2) {
                /* np.seterr(all='raise') (Pore:11): divide-by-zero / invalid -> FloatingPointError aborts the run */
                if (P->reserved1 & 1) { ORC(fp_tolerated)++; continue; }     /* option: skip the pair, keep going */
                rc = AMC_ERR_FP;
                goto done;
            }
            double sq = sqrt(disc2);
            double t1 = (-b + sq) / (2 * a), t2 = (-b - sq) / (2 * a);
            double t = (t1 > t2) ? t1 : t2;                                                  /* Pore:185 np.max */
            if (flag[j]) {                                                                   /* Pore:186-190 */
                orc_emit(sink, step, phase, cell, gidx ? gidx[i] : (int32_t)i, gidx ? gidx[j] : (int32_t)j, 0,
                         fabs(cont[j] - fabs(sqrt(SQ(vx1) + SQ(vy1) + SQ(vz1)) * t)), fabs(contx[j] - fabs(vx1 * t)),
                         fabs(conty[j] - fabs(vy1 * t)), fabs(contz[j] - fabs(vz1 * t)));
            } else {
                flag[j] = 1;                                                                 /* Pore:192 */
            }
            if (flag[i]) {                                                                   /* Pore:193-197 */
                orc_emit(sink, step, phase, cell, gidx ? gidx[i] : (int32_t)i, gidx ? gidx[j] : (int32_t)j, 1,
                         fabs(cont[i] - fabs(sqrt(SQ(vx2) + SQ(vy2) + SQ(vz2)) * t)), fabs(contx[i] - fabs(vx2 * t)),
                         fabs(conty[i] - fabs(vy2 * t)), fabs(contz[i] - fabs(vz2 * t)));
            } else {
                flag[i] = 1;                                                                 /* Pore:199 */
            }
            /* Pore:202 positions at contact */
            double nx1 = x1 - vx1 * t, ny1 = y1 - vy1 * t, nz1 = z1 - vz1 * t;
            double nx2 = x2 - vx2 * t, ny2 = y2 - vy2 * t, nz2 = z2 - vz2 * t;
            /* Pore:205-207 unit normal */
            double n0 = (nx2 - nx1) / cr, n1 = (ny2 - ny1) / cr, n2 = (nz2 - nz1) / cr;
            /* Pore:209 */
            double d1 = fma(vz1, n2, fma(vy1, n1, vx1 * n0));
            double d2 = fma(vz2, n2, fma(vy2, n1, vx2 * n0));
            double p = (d1 - d2) / m;
            double pm = p * m;
            double wvx1 = vx1 - pm * n0, wvy1 = vy1 - pm * n1, wvz1 = vz1 - pm * n2;         /* Pore:211-213 */
            double wvx2 = vx2 + pm * n0, wvy2 = vy2 + pm * n1, wvz2 = vz2 + pm * n2;         /* Pore:214-216 */
            xs[j] = nx1 + wvx1 * t; ys[j] = ny1 + wvy1 * t; zs[j] = nz1 + wvz1 * t;         /* Pore:218,221-223 */
            xs[i] = nx2 + wvx2 * t; ys[i] = ny2 + wvy2 * t; zs[i] = nz2 + wvz2 * t;         /* Pore:219,224-226 */
            vxs[j] = wvx1; vys[j] = wvy1; vzs[j] = wvz1;                                     /* Pore:227-229 */
            vxs[i] = wvx2; vys[i] = wvy2; vzs[i] = wvz2;                                     /* Pore:230-232 */
            cont[i] = fabs(sqrt(SQ(wvx2) + SQ(wvy2) + SQ(wvz2)) * t);                        /* Pore:233 */
            cont[j] = fabs(sqrt(SQ(wvx1) + SQ(wvy1) + SQ(wvz1)) * t);                        /* Pore:234 */
            contx[i] = fabs(wvx2 * t); contz[i] = fabs(wvz2 * t); conty[i] = fabs(wvy2 * t); /* Pore:235-237 */
            contx[j] = fabs(wvx1 * t); conty[j] = fabs(wvy1 * t); contz[j] = fabs(wvz1 * t); /* Pore:238-240 */
            ncoll++;                                                                         /* Pore:241 */
        }
    }
done:
    if (ncoll_out) *ncoll_out += ncoll;
    return rc;
}

/* ------------------------------------------------------------------------------------------------------- */
/* a6 — hit_vertical_wall, Pore:257-292.  hits = boolean mask over all particles.                            */
int ORC(pore_vertical_wall)(const amc_params *P, orc_state *S, const uint8_t *hits, double z_plane, orc_sink *sink,
                            int32_t step, int32_t phase, int64_t *ncoll)
{
    (void)P;
    for (int64_t p = 0; p < S->n; p++) {
        if (!hits[p]) continue;
        double t = (S->z[p] - z_plane) / S->vz[p];                                           /* Pore:261 (array op) */
        double vx = S->vx[p], vy = S->vy[p], vz = S->vz[p];
        if (S->flag[p]) {                                                                    /* Pore:274-278 */
            orc_emit(sink, step, phase, 0, (int32_t)p, -1, 0, fabs(S->d[p] - fabs(sqrt(SQ(vx) + SQ(vy) + SQ(vz)) * t)),
                     fabs(S->dx[p] - fabs(vx * t)), fabs(S->dy[p] - fabs(vy * t)), fabs(S->dz[p] - fabs(vz * t)));
        } else {
            S->flag[p] = 1;
        }
        S->d[p] = fabs(sqrt(SQ(vx) + SQ(vy) + SQ(vz)) * t);                                  /* Pore:281 */
        S->dx[p] = fabs(vx * t); S->dy[p] = fabs(vy * t); S->dz[p] = fabs(vz * t);           /* Pore:282-284 */
        S->vz[p] = -vz;                                                                      /* Pore:290 */
        S->z[p] = z_plane + t * S->vz[p];                                                    /* Pore:291 */
        (*ncoll)++;                                                                          /* Pore:292 */
    }
    return 0;
}

/* a6 — hit_cylinder_side_wall, Pore:294-348.  bookkeeping != 0 -> Pore variant (paths + counter);
 * bookkeeping == 0 -> Temp's hit_cylinder_specular_side_wall (Temp:317-347): geometry only, errors counted. */
int ORC(side_wall)(const amc_params *P, orc_state *S, const uint8_t *hits, double Rc, int bookkeeping, orc_sink *sink,
                   int32_t step, int32_t phase, int64_t *ncoll, int64_t *nerr)
{
    for (int64_t p = 0; p < S->n; p++) {
        if (!hits[p]) continue;
        double x = S->x[p], y = S->y[p], vx = S->vx[p], vy = S->vy[p], vz = S->vz[p];
        double a = SQ(-vx) + SQ(-vy);                                                        /* Pore:312 */
        double b = 2 * (x * (-vx) + y * (-vy));                                              /* Pore:313 */
        double c = SQ(x) + SQ(y) - SQ(Rc);                                                   /* Pore:314 */
        double disc2 = SQ(b) - 4 * a * c;
        if (a == 0.0 || disc2 < 0.0 || disc2 != disc2) {
            /* Pore:336-338: the except branch raises UnboundLocalError (total_errs not global) -> run aborts.
             * Temp:340-342: total_errs += 1, particle left untouched. */
            if (bookkeeping && !(P->reserved1 & 1)) return AMC_ERR_FP;
            if (nerr) (*nerr)++; else ORC(fp_tolerated)++;
            continue;
        }
        double sq = sqrt(disc2);
        double t1 = (-b + sq) / (2 * a), t2 = (-b - sq) / (2 * a);
        double t = (t1 < t2) ? t1 : t2;                                                      /* Pore:315 np.min */
        double cx = x - vx * t, cy = y - vy * t;                                             /* Pore:316 */
        double n0 = cx / Rc, n1 = cy / Rc;                                                   /* Pore:318 */
        double scalar = fma(vy, n1, vx * n0);                                                /* Pore:320 np.dot */
        double s2 = 2 * scalar;
        double wvx = vx - s2 * n0, wvy = vy - s2 * n1;                                       /* Pore:321 */
        double wx = cx + wvx * t, wy = cy + wvy * t;                                         /* Pore:323 */
        if (bookkeeping) {
            if (S->flag[p]) {                                                                /* Pore:324-328 */
                orc_emit(sink, step, phase, 0, (int32_t)p, -1, 0,
                         fabs(S->d[p] - fabs(sqrt(SQ(vx) + SQ(vy) + SQ(vz)) * t)), fabs(S->dx[p] - fabs(vx * t)),
                         fabs(S->dy[p] - fabs(vy * t)), fabs(S->dz[p] - fabs(vz * t)));
            } else {
                S->flag[p] = 1;
            }
            S->d[p] = fabs(sqrt(SQ(wvx) + SQ(wvy) + SQ(vz)) * t);                            /* Pore:332 */
            S->dx[p] = fabs(wvx * t); S->dy[p] = fabs(wvy * t); S->dz[p] = fabs(vz * t);     /* Pore:333-335 */
            (*ncoll)++;                                                                      /* Pore:348 */
        }
        S->x[p] = wx; S->y[p] = wy; S->vx[p] = wvx; S->vy[p] = wvy;                          /* Pore:331,339-342 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------- */
/* a4 — drift, Cube:179-187 / Pore:426-437 (all ndarray ops: exact products, no FMA)                          */
void ORC(drift)(const amc_params *P, orc_state *S, double dt, int save_prior)
{
    (void)P;
    for (int64_t p = 0; p < S->n; p++) {
        if (save_prior) { S->px[p] = S->x[p]; S->py[p] = S->y[p]; S->pz[p] = S->z[p]; }      /* Pore:427-429 */
        double sx = dt * S->vx[p], sy = dt * S->vy[p], sz = dt * S->vz[p];
        S->x[p] += sx; S->y[p] += sy; S->z[p] += sz;                                         /* Pore:430-432 */
        S->d[p] += fabs(sqrt(sx * sx + sy * sy + sz * sz));                                  /* Pore:434 */
        S->dx[p] += fabs(sx); S->dy[p] += fabs(sy); S->dz[p] += fabs(sz);                    /* Pore:435-437 */
    }
}

/* a5 — cube walls, Cube:189-226: per axis, max wall then min wall (ndarray ops) */
static void ORC(cube_axis)(int64_t n, double *pos, double *vel, double W)
{
    for (int64_t p = 0; p < n; p++) {
        if (pos[p] > W) {                                                                    /* Cube:192-195 */
            double dt_ac = (pos[p] - W) / vel[p];
            vel[p] = -vel[p];
            pos[p] = W + dt_ac * vel[p];
        }
        if (pos[p] < 0) {                                                                    /* Cube:197-200 */
            double dt_ac = pos[p] / vel[p];
            vel[p] = -vel[p];
            pos[p] = dt_ac * vel[p];
        }
    }
}
void ORC(cube_walls)(const amc_params *P, orc_state *S)
{
    ORC(cube_axis)(S->n, S->x, S->vx, P->cube_x);
    ORC(cube_axis)(S->n, S->y, S->vy, P->cube_y);
    ORC(cube_axis)(S->n, S->z, S->vz, P->cube_z);
}

/* a6 — the six Pore wall cases in order, Pore:439-485.  Each mask is evaluated after the previous handler ran. */
int ORC(pore_walls)(const amc_params *P, orc_state *S, orc_sink *sink, int32_t step, int64_t *nwall)
{
    const int64_t n = S->n;
    uint8_t *hits = (uint8_t *)malloc((size_t)n + 1);
    int rc = 0;
    const double zb = P->z_gap_bottom, zt = P->z_gap_top;
#define RAD(p) sqrt(S->x[p] * S->x[p] + S->y[p] * S->y[p])
#define RAD0(p) sqrt(S->px[p] * S->px[p] + S->py[p] * S->py[p])
    /* CASE 1, Pore:442-443 */
    for (int64_t p = 0; p < n; p++) hits[p] = RAD(p) > P->R_oa;
    if ((rc = ORC(side_wall)(P, S, hits, P->R_oa_c, 1, sink, step, 1, nwall, NULL))) goto out;
    /* CASE 2, Pore:448-452 */
    for (int64_t p = 0; p < n; p++) hits[p] = S->z[p] < 0;
    ORC(pore_vertical_wall)(P, S, hits, 0.0, sink, step, 2, nwall);
    for (int64_t p = 0; p < n; p++) hits[p] = S->z[p] > P->H;
    ORC(pore_vertical_wall)(P, S, hits, P->H, sink, step, 3, nwall);
    /* CASE 3, Pore:457-461 */
    for (int64_t p = 0; p < n; p++) hits[p] = (S->pz[p] > P->z_cold) && (S->z[p] < P->z_cold) && (RAD(p) > P->R_p);
    ORC(pore_vertical_wall)(P, S, hits, P->z_cold, sink, step, 4, nwall);
    for (int64_t p = 0; p < n; p++) hits[p] = (S->pz[p] < P->h_oa) && (S->z[p] > P->h_oa) && (RAD(p) > P->R_p);
    ORC(pore_vertical_wall)(P, S, hits, P->h_oa, sink, step, 5, nwall);
    /* CASE 4, Pore:465-467 */
    for (int64_t p = 0; p < n; p++)
        hits[p] = (S->pz[p] < zt) && (S->pz[p] > zb) && (RAD0(p) < P->R_g) && (RAD(p) > P->R_g);
    if ((rc = ORC(side_wall)(P, S, hits, P->R_g_c, 1, sink, step, 6, nwall, NULL))) goto out;
    /* CASE 5, Pore:472-478 */
    for (int64_t p = 0; p < n; p++)
        hits[p] = (RAD0(p) > P->R_p) && (S->z[p] < zb) && (S->pz[p] < zt) && (S->pz[p] > zb);
    ORC(pore_vertical_wall)(P, S, hits, zb, sink, step, 7, nwall);
    for (int64_t p = 0; p < n; p++)
        hits[p] = (RAD0(p) > P->R_p) && (S->z[p] > zt) && (S->pz[p] < zt) && (S->pz[p] > zb);
    ORC(pore_vertical_wall)(P, S, hits, zt, sink, step, 8, nwall);
    /* CASE 6, Pore:482-485 */
    for (int64_t p = 0; p < n; p++)
        hits[p] = (RAD0(p) < P->R_p) && (RAD(p) > P->R_p) &&
                  (((S->z[p] < P->z_cold) && (S->z[p] > zt)) || ((S->z[p] < zb) && (S->z[p] > P->h_oa)));
    if ((rc = ORC(side_wall)(P, S, hits, P->R_p_c, 1, sink, step, 9, nwall, NULL))) goto out;
#undef RAD
#undef RAD0
out:
    free(hits);
    return rc;
}

/* a8 — num_out_of_bounds, Pore:354-375 (MUTATES: it is called from inside print(), Pore:512,550), and
 * recapture_out_of_bounds, Temp:594-616.  energised selects Temp's assignment form for the two z tests. */
int64_t ORC(bounds)(const amc_params *P, orc_state *S, int energised)
{
    int64_t cnt = 0;
    for (int64_t p = 0; p < S->n; p++) {
        /* the five tests run array-wide one after another in the reference; each touches only particle p and
         * reads p's current values, so running all five per particle is equivalent */
        if (S->z[p] < 0) { if (energised) S->z[p] = P->oob_z_lo_fix; else S->z[p] += P->oob_z_lo_fix; cnt++; } /* 357-359 */
        if (S->z[p] > P->H) { if (energised) S->z[p] = P->oob_z_hi_fix; else S->z[p] -= P->oob_z_hi_fix; cnt++; } /* 360-362 */
        if (S->x[p] * S->x[p] + S->y[p] * S->y[p] > P->R_oa_sq) { S->x[p] = 0; S->y[p] = 0; cnt++; }          /* 363-366 */
        if ((S->x[p] * S->x[p] + S->y[p] * S->y[p] > P->R_g_sq) && (S->z[p] > P->h_oa) && (S->z[p] < P->z_cold)) {
            S->x[p] = 0; S->y[p] = 0; cnt++;                                                                  /* 367-370 */
        }
        if ((S->x[p] * S->x[p] + S->y[p] * S->y[p] > P->R_p_sq) &&
            (((S->z[p] > P->h_oa) && (S->z[p] < P->z_oob_hot_top)) || ((S->z[p] > P->z_oob_gap_top) && (S->z[p] < P->z_cold)))) {
            S->x[p] = 0; S->y[p] = 0; cnt++;                                                                  /* 371-374 */
        }
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------------------- */
/* a2 — p-p driver, Pore:520-549 == Temp:813-842.  8 colour groups; in each, every particle is a member of at
 * most one cell; members are gathered in ascending index, cells run in (lx,ly,lz) list order.              */
static int ORC(axis_cell)(double v, int grp, int nsub_layers, int offset, double d, double ov)
{
    /* find layer l in [0,nsub_layers) with ((2l+grp-offset)*d - ov) < v  &&  v < ((2l+grp-offset+1)*d)
     * (Pore:527-529; offset = num_subdivions for x,y and 0 for z) */
    double f = floor(v / d);
    for (int dk = -1; dk <= 1; dk++) {
        long k = (long)f + dk;               /* k = 2l+grp-offset */
        long twol = k - grp + offset;
        if (twol < 0 || (twol & 1)) continue;
        long l = twol / 2;
        if (l >= nsub_layers) continue;
        double lo = (double)k * d - ov, hi = (double)(k + 1) * d;
        if (lo < v && v < hi) return (int)l;
    }
    return -1;
}

typedef struct { int64_t cell; int32_t idx; } ORC(memb);

static int ORC(run_cell)(const amc_params *P, orc_state *S, const ORC(memb) *mem, int64_t cnt, orc_sink *sink,
                         int32_t step, int32_t phase, int64_t cell, int64_t *npp, double *buf, uint8_t *fbuf, int32_t *ibuf)
{
    double *cont = buf, *cx = buf + cnt, *cy = buf + 2 * cnt, *cz = buf + 3 * cnt, *x = buf + 4 * cnt,
           *y = buf + 5 * cnt, *z = buf + 6 * cnt, *vx = buf + 7 * cnt, *vy = buf + 8 * cnt, *vz = buf + 9 * cnt;
    for (int64_t k = 0; k < cnt; k++) {                                                      /* Pore:533-543 */
        int32_t p = mem[k].idx;
        ibuf[k] = p; cont[k] = S->d[p]; cx[k] = S->dx[p]; cy[k] = S->dy[p]; cz[k] = S->dz[p]; fbuf[k] = S->flag[p];
        x[k] = S->x[p]; y[k] = S->y[p]; z[k] = S->z[p]; vx[k] = S->vx[p]; vy[k] = S->vy[p]; vz[k] = S->vz[p];
    }
    int rc = ORC(pair_cell)(P, cnt, cont, cx, cy, cz, fbuf, x, y, z, vx, vy, vz, ibuf, sink, step, phase, cell, npp);
    for (int64_t k = 0; k < cnt; k++) {                                                      /* Pore:547 */
        int32_t p = ibuf[k];
        S->d[p] = cont[k]; S->dx[p] = cx[k]; S->dy[p] = cy[k]; S->dz[p] = cz[k]; S->flag[p] = fbuf[k];
        S->x[p] = x[k]; S->y[p] = y[k]; S->z[p] = z[k]; S->vx[p] = vx[k]; S->vy[p] = vy[k]; S->vz[p] = vz[k];
    }
    return rc;
}

int ORC(pore_sweep)(const amc_params *P, orc_state *S, orc_sink *sink, int32_t step, int64_t *npp, int64_t *npairs_tested)
{
    const int64_t n = S->n;
    ORC(memb) *mem = (ORC(memb) *)malloc(sizeof(ORC(memb)) * (size_t)(n + 1));
    double *buf = (double *)malloc(sizeof(double) * 10 * (size_t)(n + 1));
    uint8_t *fbuf = (uint8_t *)malloc((size_t)n + 1);
    int32_t *ibuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int64_t *pcell = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *cstart = (int64_t *)malloc(sizeof(int64_t) * (size_t)((int64_t)P->nx * P->ny * (P->nz / 2) + 2));
    int rc = 0;
    for (int gx = 0; gx < 2 && !rc; gx++)
        for (int gy = 0; gy < 2 && !rc; gy++)
            for (int gz = 0; gz < 2 && !rc; gz++) {                                          /* Pore:522-524 */
                /* stable counting sort by cell: members of a cell end up in ascending particle index (Pore:538) */
                const int64_t ncell = (int64_t)P->nx * P->ny * (P->nz / 2);
                memset(cstart, 0, sizeof(int64_t) * (size_t)(ncell + 1));
                for (int64_t p = 0; p < n; p++) {
                    pcell[p] = -1;
                    int lx = ORC(axis_cell)(S->x[p], gx, P->nx, P->nx, P->dx, P->overlap_x);
                    if (lx < 0) continue;
                    int ly = ORC(axis_cell)(S->y[p], gy, P->ny, P->ny, P->dy, P->overlap_y);
                    if (ly < 0) continue;
                    int lz = ORC(axis_cell)(S->z[p], gz, P->nz / 2, 0, P->dz, P->overlap_z);
                    if (lz < 0) continue;
                    pcell[p] = ((int64_t)lx * P->ny + ly) * (P->nz / 2) + lz;               /* Pore:530 list order */
                    cstart[pcell[p] + 1]++;
                }
                for (int64_t c = 0; c < ncell; c++) cstart[c + 1] += cstart[c];
                int64_t cnt = cstart[ncell];
                for (int64_t p = 0; p < n; p++) {
                    if (pcell[p] < 0) continue;
                    int64_t k = cstart[pcell[p]]++;
                    mem[k].cell = pcell[p];
                    mem[k].idx = (int32_t)p;
                }
                int phase = 16 + 4 * gx + 2 * gy + gz;
                for (int64_t s = 0; s < cnt && !rc;) {
                    int64_t e = s;
                    while (e < cnt && mem[e].cell == mem[s].cell) e++;
                    if (npairs_tested) *npairs_tested += (e - s) * (e - s - 1) / 2;
                    rc = ORC(run_cell)(P, S, mem + s, e - s, sink, step, phase, mem[s].cell, npp, buf, fbuf, ibuf);
                    s = e;
                }
            }
    free(mem); free(buf); free(fbuf); free(ibuf); free(pcell); free(cstart);
    return rc;
}

/* ---- all host cores: the reference's own parallel structure (Pore:520-549: the cells of one colour group are disjoint
 * and go to a pool of workers, the eight groups follow each other), with OpenMP threads in place of the Pool's processes.
 * Inside a group the order of the cells only decides the order in which completed paths are appended, not the state, so
 * the result equals ORC(pore_sweep)'s (tests/test_oracle_parallel.py); completed paths are counted, not recorded.
 * cube != 0 applies the same colouring to the Cube script's cells (l*d - ov < v < (l+1)*d, colour = parity of l): the
 * reference's cube loop is SERIAL (Cube:231-336, lexicographic with stale masks), so for that geometry this is what all
 * cores could do with the Pore script's scheme — a timing baseline with a different processing order, not a port.  */
static int ORC(axis_cell_cube)(double v, int grp, int nlayers, double d, double ov)
{
    double f = floor(v / d);
    for (int dk = 0; dk <= 1; dk++) {
        long l = (long)f + dk;
        if (l < 0 || l >= nlayers || ((l & 1) != grp)) continue;
        double lo = (double)l * d - ov, hi = (double)(l + 1) * d;
        if (lo < v && v < hi) return (int)(l >> 1);
    }
    return -1;
}

int ORC(sweep_par)(const amc_params *P, orc_state *S, int cube, int64_t *npp, int64_t *npaths, int64_t *npairs_tested)
{
    const int64_t n = S->n;
    const int hx = cube ? (P->nx + 1) / 2 : P->nx, hy = cube ? (P->ny + 1) / 2 : P->ny, hz = cube ? (P->nz + 1) / 2 : P->nz / 2;
    const int64_t ncell = (int64_t)hx * hy * hz;
    ORC(memb) *mem = (ORC(memb) *)malloc(sizeof(ORC(memb)) * (size_t)(n + 1));
    int64_t *pcell = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *cstart = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncell + 2));
    int64_t *seg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncell + 2));
    int rc_all = 0;
    int64_t tot_pp = 0, tot_paths = 0, tot_pairs = 0;
    for (int gx = 0; gx < 2; gx++)
        for (int gy = 0; gy < 2; gy++)
            for (int gz = 0; gz < 2; gz++) {
                memset(cstart, 0, sizeof(int64_t) * (size_t)(ncell + 1));
#pragma omp parallel for schedule(static)
                for (int64_t p = 0; p < n; p++) {
                    pcell[p] = -1;
                    int lx = cube ? ORC(axis_cell_cube)(S->x[p], gx, P->nx, P->dx, P->overlap_x) : ORC(axis_cell)(S->x[p], gx, P->nx, P->nx, P->dx, P->overlap_x);
                    if (lx < 0) continue;
                    int ly = cube ? ORC(axis_cell_cube)(S->y[p], gy, P->ny, P->dy, P->overlap_y) : ORC(axis_cell)(S->y[p], gy, P->ny, P->ny, P->dy, P->overlap_y);
                    if (ly < 0) continue;
                    int lz = cube ? ORC(axis_cell_cube)(S->z[p], gz, P->nz, P->dz, P->overlap_z) : ORC(axis_cell)(S->z[p], gz, P->nz / 2, 0, P->dz, P->overlap_z);
                    if (lz < 0) continue;
                    pcell[p] = ((int64_t)lx * hy + ly) * hz + lz;
                }
                for (int64_t p = 0; p < n; p++)
                    if (pcell[p] >= 0) cstart[pcell[p] + 1]++;
                for (int64_t c = 0; c < ncell; c++) cstart[c + 1] += cstart[c];
                const int64_t cnt = cstart[ncell];
                for (int64_t c = 0; c <= ncell; c++) seg[c] = cstart[c];
                for (int64_t p = 0; p < n; p++) {                 /* stable: members of a cell in ascending particle index */
                    if (pcell[p] < 0) continue;
                    int64_t k = cstart[pcell[p]]++;
                    mem[k].cell = pcell[p];
                    mem[k].idx = (int32_t)p;
                }
                (void)cnt;
                const int phase = 16 + 4 * gx + 2 * gy + gz;
#pragma omp parallel reduction(+ : tot_pp, tot_paths, tot_pairs) reduction(| : rc_all)
                {
                    int64_t cap = 64;
                    double *buf = (double *)malloc(sizeof(double) * 10 * (size_t)cap);
                    uint8_t *fbuf = (uint8_t *)malloc((size_t)cap);
                    int32_t *ibuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
                    amc_path_record recs[8];
#pragma omp for schedule(dynamic, 16)
                    for (int64_t c = 0; c < ncell; c++) {
                        const int64_t s = seg[c], m = seg[c + 1] - seg[c];
                        if (m < 2) continue;
                        if (m > cap) {
                            cap = 2 * m;
                            free(buf); free(fbuf); free(ibuf);
                            buf = (double *)malloc(sizeof(double) * 10 * (size_t)cap);
                            fbuf = (uint8_t *)malloc((size_t)cap);
                            ibuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
                        }
                        orc_sink sink = {recs, 0, 0, 0};            /* capacity 0: every completed path counts as overflow */
                        int64_t pp = 0;
                        rc_all |= ORC(run_cell)(P, S, mem + s, m, &sink, 0, phase, c, &pp, buf, fbuf, ibuf);
                        tot_pp += pp;
                        tot_paths += sink.overflow;
                        tot_pairs += m * (m - 1) / 2;
                    }
                    free(buf); free(fbuf); free(ibuf);
                }
            }
    free(mem); free(pcell); free(cstart); free(seg);
    if (npp) *npp += tot_pp;
    if (npaths) *npaths += tot_paths;
    if (npairs_tested) *npairs_tested += tot_pairs;
    return rc_all;
}

/* one step with the parallel sweep (specular geometries): per-particle stages as in ORC(timestep) */
int ORC(timestep_par)(const amc_params *P, orc_state *S, double dt, amc_step_stats *st)
{
    amc_step_stats z;
    memset(&z, 0, sizeof z);
    int rc = 0;
    const int cube = P->geometry == AMC_GEOM_CUBE;
    ORC(drift)(P, S, dt, !cube);
    if (cube) {
        ORC(cube_walls)(P, S);
    } else {
        orc_sink none = {0, 0, 0, 0};
        rc = ORC(pore_walls)(P, S, &none, 0, &z.n_wall);
        z.n_paths += none.overflow;
        if (!rc) z.n_oob_walls = ORC(bounds)(P, S, 0);
    }
    if (!rc) rc = ORC(sweep_par)(P, S, cube, &z.n_pp, &z.n_paths, &z.n_candidates);
    if (!rc && !cube) z.n_oob_pp = ORC(bounds)(P, S, 0);
    if (st) *st = z;
    return rc;
}

/* a3 — Cube cell loop, Cube:231-336: lexicographic (x,y,z) cells; in_x_layer is evaluated once per x_layer,
 * in_y_layer once per (x_layer,y_layer), in_z_layer per cell — all from the state at that moment; every cell
 * gathers from / scatters to the global arrays before the next one runs.                                     */
int ORC(cube_sweep)(const amc_params *P, orc_state *S, orc_sink *sink, int32_t step, int64_t *npp, int64_t *npairs_tested)
{
    const int64_t n = S->n;
    int32_t *Lx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t *Lxy = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    ORC(memb) *mem = (ORC(memb) *)malloc(sizeof(ORC(memb)) * (size_t)(n + 1));
    double *buf = (double *)malloc(sizeof(double) * 10 * (size_t)(n + 1));
    uint8_t *fbuf = (uint8_t *)malloc((size_t)n + 1);
    int32_t *ibuf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int rc = 0;
    for (int lx = 0; lx < P->nx && !rc; lx++) {                                              /* Cube:232 */
        double xlo = lx * P->dx - P->overlap_x, xhi = (lx + 1) * P->dx;                      /* Cube:233 */
        int64_t nLx = 0;
        for (int64_t p = 0; p < n; p++)
            if (xlo < S->x[p] && S->x[p] < xhi) Lx[nLx++] = (int32_t)p;
        for (int ly = 0; ly < P->ny && !rc; ly++) {                                          /* Cube:234 */
            double ylo = ly * P->dy - P->overlap_y, yhi = (ly + 1) * P->dy;                  /* Cube:235 */
            int64_t nLxy = 0;
            for (int64_t k = 0; k < nLx; k++) {
                int32_t p = Lx[k];
                if (ylo < S->y[p] && S->y[p] < yhi) Lxy[nLxy++] = p;
            }
            for (int lz = 0; lz < P->nz && !rc; lz++) {                                      /* Cube:236 */
                double zlo = lz * P->dz - P->overlap_z, zhi = (lz + 1) * P->dz;              /* Cube:237 */
                int64_t cnt = 0;
                int64_t cell = ((int64_t)lx * P->ny + ly) * P->nz + lz;
                for (int64_t k = 0; k < nLxy; k++) {
                    int32_t p = Lxy[k];
                    if (zlo < S->z[p] && S->z[p] < zhi) { mem[cnt].cell = cell; mem[cnt].idx = p; cnt++; }
                }
                if (cnt < 2) continue;
                if (npairs_tested) *npairs_tested += cnt * (cnt - 1) / 2;
                rc = ORC(run_cell)(P, S, mem, cnt, sink, step, 16, cell, npp, buf, fbuf, ibuf);
            }
        }
    }
    free(Lx); free(Lxy); free(mem); free(buf); free(fbuf); free(ibuf);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------- */
/* a7 — Temperature_Pore_MC.py walls.  Cases 1-2 are specular WITHOUT bookkeeping (Temp:311-347); cases 3-6 re-emit
 * the particle in a random direction with an accommodated energy (Temp:349-553).  The random directions
 * (Temp:132-141: np.random + random Mersenne Twisters) and the gap-wall Debye energy (mpmath.quad, Temp:147-152) are
 * drawn by the HOST in ascending particle index and handed in; everything deterministic is restated here.
 * case ids in evaluation order: 3 = case-3 cold plane, 4 = case-3 hot plane, 5 = case-4 gap side wall,
 * 6 = case-5 bottom plane, 7 = case-5 top plane, 8 = case-6 hot side wall, 9 = case-6 cold side wall.            */
void ORC(temp_specular)(const amc_params *P, orc_state *S, int64_t *nerr)
{
    const int64_t n = S->n;
    uint8_t *hits = (uint8_t *)malloc((size_t)n + 1);
    for (int64_t p = 0; p < n; p++) hits[p] = sqrt(S->x[p] * S->x[p] + S->y[p] * S->y[p]) > P->R_oa;   /* Temp:693 */
    ORC(side_wall)(P, S, hits, P->R_oa_c, 0, NULL, 0, 1, NULL, nerr);                                   /* Temp:694 */
    for (int64_t p = 0; p < n; p++) {
        if (S->z[p] < 0) {                                                                   /* Temp:699-700, 311-315 */
            double t = (S->z[p] - 0.0) / S->vz[p];
            S->vz[p] = -S->vz[p];
            S->z[p] = 0.0 + t * S->vz[p];
        }
        if (S->z[p] > P->H) {                                                                /* Temp:702-703 */
            double t = (S->z[p] - P->H) / S->vz[p];
            S->vz[p] = -S->vz[p];
            S->z[p] = P->H + t * S->vz[p];
        }
    }
    free(hits);
}

void ORC(temp_mask)(const amc_params *P, const orc_state *S, int case_id, uint8_t *hits)
{
    for (int64_t p = 0; p < S->n; p++) {
        const double x = S->x[p], y = S->y[p], z = S->z[p], px = S->px[p], py = S->py[p], pz = S->pz[p];
        const double r2 = x * x + y * y, r02 = px * px + py * py;
        int h = 0;
        switch (case_id) {
        case 3: h = (pz >= P->t_z3_cold) && (z < P->t_z3_cold) && (r2 > P->R_p_sq); break;               /* Temp:708 */
        case 4: h = (pz <= P->t_z3_hot) && (z > P->t_z3_hot) && (r2 > P->R_p_sq); break;                 /* Temp:713 */
        case 5: h = (pz < P->t_zgap_hi) && (pz > P->t_zgap_lo) && (r02 <= P->R_g_c_sq) && (r2 > P->R_g_c_sq); break; /* 720 */
        case 6: h = (r02 >= P->R_p_c_sq) && (z < P->t_zgap_lo) && (pz <= P->t_zgap_hi) && (pz >= P->t_zgap_lo); break; /* 728 */
        case 7: h = (r02 >= P->R_p_c_sq) && (z > P->t_zgap_hi) && (pz <= P->t_zgap_hi) && (pz >= P->t_zgap_lo); break; /* 734 */
        case 8: h = (r02 <= P->R_p_c_sq) && (r2 > P->R_p_c_sq) && (z <= P->t_zgap_lo) && (z >= P->t_z3_hot); break;   /* 743 */
        case 9: h = (r02 <= P->R_p_c_sq) && (r2 > P->R_p_c_sq) && (z < P->t_z3_cold) && (z > P->t_zgap_hi); break;   /* 749 */
        default: break;
        }
        hits[p] = (uint8_t)h;
    }
}

static int ORC(temp_is_plane)(int case_id) { return case_id == 3 || case_id == 4 || case_id == 6 || case_id == 7; }
static double ORC(temp_plane)(const amc_params *P, int case_id)
{
    return case_id == 3 ? P->t_z3_cold : case_id == 4 ? P->t_z3_hot : case_id == 6 ? P->t_zgap_lo : P->t_zgap_hi;
}
static double ORC(temp_radius)(const amc_params *P, int case_id) { return case_id == 5 ? P->R_g_c : P->R_p_c; }

/* geometry of the hits of one case, in ascending particle index: flight time since contact, contact point, the
 * INWARD unit normal handed to random_inbounds_direction (Temp:374-375, 442-444) and ok = 0 where the reference's
 * try-block fails (Temp:472-474).  Returns the number of hits. */
int64_t ORC(temp_geometry)(const amc_params *P, const orc_state *S, int case_id, const uint8_t *hits, int32_t *idx,
                           double *t_out, double *contact, double *normal, uint8_t *ok)
{
    int64_t k = 0;
    for (int64_t p = 0; p < S->n; p++) {
        if (!hits[p]) continue;
        const double x = S->x[p], y = S->y[p], z = S->z[p], vx = S->vx[p], vy = S->vy[p], vz = S->vz[p];
        idx[k] = (int32_t)p;
        ok[k] = 1;
        if (ORC(temp_is_plane)(case_id)) {
            const double zp = ORC(temp_plane)(P, case_id);
            const double t = (z - zp) / vz;                                                  /* Temp:353 */
            t_out[k] = t;
            contact[3 * k] = x - vx * t; contact[3 * k + 1] = y - vy * t; contact[3 * k + 2] = zp;   /* Temp:372 */
            normal[3 * k] = 0; normal[3 * k + 1] = 0;
            normal[3 * k + 2] = (case_id == 3 || case_id == 6) ? 1.0 : -1.0;                 /* Temp:709,714,730,736 */
        } else {
            const double Rc = ORC(temp_radius)(P, case_id);
            const double a = SQ(-vx) + SQ(-vy);                                              /* Temp:436 */
            const double b = 2 * (x * (-vx) + y * (-vy));
            const double c = SQ(x) + SQ(y) - SQ(Rc);
            const double disc2 = SQ(b) - 4 * a * c;
            if (a == 0.0 || disc2 < 0.0 || disc2 != disc2) {
                ok[k] = 0; t_out[k] = 0;
                contact[3 * k] = contact[3 * k + 1] = contact[3 * k + 2] = 0;
                normal[3 * k] = normal[3 * k + 1] = normal[3 * k + 2] = 0;
            } else {
                const double sq = sqrt(disc2);
                const double t1 = (-b + sq) / (2 * a), t2 = (-b - sq) / (2 * a);
                const double t = (t1 < t2) ? t1 : t2;                                        /* Temp:439 */
                t_out[k] = t;
                const double cx = x - vx * t, cy = y - vy * t, cz = z - vz * t;              /* Temp:440 */
                contact[3 * k] = cx; contact[3 * k + 1] = cy; contact[3 * k + 2] = cz;
                /* normalized_norm_vect = [col_x, col_y, 0] / Rc ; the sampler gets its negation (Temp:442-444) */
                normal[3 * k] = -(cx / Rc); normal[3 * k + 1] = -(cy / Rc); normal[3 * k + 2] = -(0.0 / Rc);
            }
        }
        k++;
    }
    return k;
}

/* re-emission for the hits of one case (Temp:377-403, 446-471, 516-541): dir = unit direction per hit, Es = surface
 * energy per hit; writes the per-hit z-momentum and energy changes (summed by the caller in hit order, as the
 * reference accumulates them) and returns the number of hits counted into num_collisions_per_step. */
int64_t ORC(temp_apply)(const amc_params *P, orc_state *S, int case_id, int64_t nh, const int32_t *idx, const double *t_in,
                        const double *contact, const uint8_t *ok, const double *dir, const double *Es, double *dpz,
                        double *dE, orc_sink *sink, int32_t step, int64_t *nerr)
{
    const double m = P->argon_mass;
    const double alpha = (case_id == 5) ? P->alpha_gap : P->alpha_coated;
    for (int64_t k = 0; k < nh; k++) {
        const int64_t p = idx[k];
        dpz[k] = 0; dE[k] = 0;
        if (!ok[k]) { if (nerr) (*nerr)++; continue; }                                       /* Temp:472-474 */
        const double t = t_in[k];
        const double vx = S->vx[p], vy = S->vy[p], vz = S->vz[p];
        const double v_magnitude = sqrt(SQ(vx) + SQ(vy) + SQ(vz));                           /* Temp:377 */
        const double old_pz = m * vz;                                                        /* Temp:378 */
        const double E = 0.5 * m * SQ(v_magnitude);                                          /* Temp:128-129,379 */
        const double diff = Es[k] - E;                                                       /* Temp:380 */
        const double Enew = E + diff * alpha;                                                /* Temp:381 */
        const double mag = sqrt(Enew * 2 / m);                                               /* Temp:383 */
        dE[k] = Enew - E;                                                                    /* Temp:384 */
        const double wvx = dir[3 * k] * mag, wvy = dir[3 * k + 1] * mag, wvz = dir[3 * k + 2] * mag;   /* Temp:386 */
        dpz[k] = m * wvz - old_pz;                                                           /* Temp:387-388 */
        if (S->flag[p]) {                                                                    /* Temp:391-395 */
            orc_emit(sink, step, case_id + 1, 0, (int32_t)p, -1, 0,
                     fabs(S->d[p] - fabs(sqrt(SQ(vx) + SQ(vy) + SQ(vz)) * t)), fabs(S->dx[p] - fabs(vx * t)),
                     fabs(S->dy[p] - fabs(vy * t)), fabs(S->dz[p] - fabs(vz * t)));
        } else {
            S->flag[p] = 1;
        }
        S->d[p] = 0; S->dx[p] = 0; S->dy[p] = 0; S->dz[p] = 0;                               /* Temp:398-401 */
        S->x[p] = contact[3 * k]; S->y[p] = contact[3 * k + 1]; S->z[p] = contact[3 * k + 2];   /* Temp:402 */
        S->vx[p] = wvx; S->vy[p] = wvy; S->vz[p] = wvz;                                      /* Temp:403 */
    }
    return nh;                                                                               /* Temp:411,482,552 */
}

/* ------------------------------------------------------------------------------------------------------- */
/* one iteration of the reference's time loop: Cube:175-338 / Pore:416-557 (Temp's deterministic part)       */
int ORC(timestep)(const amc_params *P, orc_state *S, double dt, orc_sink *sink, int32_t step, amc_step_stats *st)
{
    int rc = 0;
    amc_step_stats z;
    memset(&z, 0, sizeof z);
    int64_t before = sink ? sink->n : 0;
    ORC(fp_tolerated) = 0;
    if (P->geometry == AMC_GEOM_CUBE) {
        ORC(drift)(P, S, dt, 0);
        ORC(cube_walls)(P, S);
        rc = ORC(cube_sweep)(P, S, sink, step, &z.n_pp, &z.n_candidates);
    } else if (P->geometry == AMC_GEOM_PORE) {
        ORC(drift)(P, S, dt, 1);
        rc = ORC(pore_walls)(P, S, sink, step, &z.n_wall);
        if (!rc) {
            z.n_oob_walls = ORC(bounds)(P, S, 0);                                            /* Pore:512 */
            rc = ORC(pore_sweep)(P, S, sink, step, &z.n_pp, &z.n_candidates);
            if (!rc) z.n_oob_pp = ORC(bounds)(P, S, 0);                                      /* Pore:550 */
        }
    } else if (P->geometry == AMC_GEOM_CELL) {
        rc = ORC(pair_cell)(P, S->n, S->d, S->dx, S->dy, S->dz, S->flag, S->x, S->y, S->z, S->vx, S->vy, S->vz, NULL,
                            sink, step, 16, 0, &z.n_pp);
    } else {
        rc = AMC_ERR_INVALID;
    }
    z.n_paths = sink ? sink->n - before : 0;
    z.n_fp_errors = ORC(fp_tolerated) + (rc == AMC_ERR_FP ? 1 : 0);
    if (st) *st = z;
    return rc;
}
