// g2g_kernels_v7.hip -- strip kernel of the record type WITHOUT gap state: Fwd2c<DPunit> (alignment mode NGP_ALB: no group
// has an internal gap), Noll 2/3.  Same recurrence and arithmetic order as g2g_forward_kernel (reference src/fwd2c.h:359-482,
// gapopen / update of DPunit src/fwd2c.cc:52-102): the gap-open cost is closed form -- Basic_GOP x (residue weight of one
// column) x (presence weight of the other), zero when the source record already runs in that direction -- so a record is
// three scalars {val, dir} that never touch memory inside a strip.
//
// Mapping = the one of g2g_kernels_v3.hip / v6 in sweep mode: a strip of 64 rows belongs to ONE wave, lane t owns row m0 + t
// and runs one column behind lane t-1; records travel down the lanes by DPP; the strip's last row goes to HBM (rowH / rowG
// / rowG2, 16 bytes per record) for the strip below, which follows on a progress counter; strips make their own column
// scores (SimBlk).  No LDS beyond the queue scratch and the staging scalars, ~60 VGPRs: as many waves as the SIMDs take.
// v1 ran these DPs with one workgroup per DP and every record in HBM.
#include <hip/hip_runtime.h>

// the two boundary chains of Fwd2c::initB (src/fwd2c.h:138-176) for DPunit: one lane walks, 16-byte records to HBM
__device__ __forceinline__ void v7_chain_tile(const DevProb &Pmem, const int which, int *prog, const int pgen)
{
    if (threadIdx.x != 0) return;
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int penc = (pgen & 0x7FF) << 20;
    double val = 0;
    int dir = D_DIAG;
    if (which == -1) {                                     // top row: corners (a.left, n), n = b.left .. a.left + rr
        unsigned *rowH = (unsigned *) P.v2_rowH + 2 * (size_t) P.v2_rowstride * 4;
        int rrt = b.right - a.left; if (P.up < rrt) rrt = P.up;
        const int nlast = a.left + rrt, ai = a.left - 1;
        { unsigned *r = rowH + (size_t) b.left * 4; *(double *) r = 0; r[2] = D_DIAG; r[3] = 0; }
        for (int n = b.left + 1; n <= nlast; ++n) {
            const int bi = n - 1;
            const double pub = unpb(P, bi, ai);
            double gnp = ishori(dir) ? 0. : P.basic_gop * (thk_at(b, bi)[0] * thk_at(a, ai)[2]);   // gapopen(prv, -1)
            gnp = (n - b.left < P.codonk1) ? gnp + pub : (P.v2divv1 * gnp + P.u2divu1 * pub);
            dir = isvert(dir) ? D_NEWH : D_HORI;
            val = val + gnp;
            unsigned *r = rowH + (size_t) n * 4;
            *(double *) r = val; r[2] = (unsigned) dir; r[3] = 0;
            if (((n - b.left) & 63) == 0) chain_publish(prog, penc, n);
        }
    } else {                                               // left column: corners (m, b.left), m = a.left .. b.left - rr
        unsigned *colH = (unsigned *) P.v2_colH;
        int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
        const int mlast = b.left - rrl, bi = b.left - 1;
        for (int m = a.left + 1; m <= mlast; ++m) {
            const int ai = m - 1;
            const double pua = unpa(P, ai, bi);
            double gnp = isvert(dir) ? 0. : P.basic_gop * (thk_at(a, ai)[0] * thk_at(b, bi)[2]);   // gapopen(prv, 1)
            gnp = (m - a.left < P.codonk1) ? gnp + pua : (P.v2divv1 * gnp + P.u2divu1 * pua);
            dir = ishori(dir) ? D_NEWV : D_VERT;
            val = val + gnp;
            unsigned *r = colH + (size_t) (m - a.left) * 4;
            *(double *) r = val; r[2] = (unsigned) dir; r[3] = 0;
            if (((m - a.left) & 63) == 0) chain_publish(prog, penc, m - a.left);
        }
    }
    chain_publish(prog, penc, 0xFFFFF);
}

template <bool NOLL3>
__device__ __forceinline__ void v7_strip(const DevProb &Pmem, lchar *lds, const int ti, const int nsteps,
                                         const int *prog_up, int *prog_self, int *dbg, const int pgen, const int pint, const int *prog_left,
                                         double *simscr, int *failp)
{
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int lane = threadIdx.x;                          // blockDim.x == 64
    lu32 *const stsc = (lu32 *) lds;                       // staging scalars of lane 0's upper neighbours: H ring 0-2, G 3-4, G2 5-6
    const size_t rbuf = (size_t) P.v2_rowstride * 4;
    const int bprev = (ti + 2) % 3, bcur = ti % 3;
    const GLB unsigned *rowHp = glb((const unsigned *) P.v2_rowH + bprev * rbuf), *rowGp = glb((const unsigned *) P.v2_rowG + bprev * rbuf);
    const GLB unsigned *rowG2p = NOLL3 ? glb((const unsigned *) P.v2_rowG2 + bprev * rbuf) : 0;
    GLB unsigned *rowHc = glbw((unsigned *) P.v2_rowH + bcur * rbuf), *rowGc = glbw((unsigned *) P.v2_rowG + bcur * rbuf);
    GLB unsigned *rowG2c = NOLL3 ? glbw((unsigned *) P.v2_rowG2 + bcur * rbuf) : 0;
    const GLB unsigned *colH = glb((const unsigned *) P.v2_colH);
    const GLB double *bthk = glb(b.thk);
    GLB uint8_t *const trace = glbw(P.trace);
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
    const int m_left_last = b.left - rrl;                  // last row whose corner (m, b.left) exists
    const int m0 = a.left + ti * 64, m = m0 + lane;
    if (prog_left) {
        const int rows_ = m0 + 64 - a.left;
        const int wantl = ((pgen & 0x7FF) << 20) | (rows_ < 0xFFFFF ? rows_ : 0xFFFFF);
        (void) g2g_wait_ge(prog_left, wantl, dbg, failp, ti);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int mend = (m0 + 64 < a.right) ? m0 + 64 : a.right;
    const int llast = mend - 1 - m0;                       // lane of the strip's last row
    const int c1 = b.right;
    const bool row_ok = m < a.right;
    int nlo = m + P.lw; if (nlo < b.left) nlo = b.left;    // the row's range, fwd2c.h:373-374
    int nhi = m + P.up + 1; if (nhi > b.right) nhi = b.right;
    const int lo = nlo, hi = nhi;
    int cbase = m0 + P.lw; if (cbase < b.left) cbase = b.left;
    int hi0 = m0 + P.up + 1; if (hi0 > b.right) hi0 = b.right;            // lane 0's hi
    const bool vert0 = m0 > a.left;                        // the strip has a row above
    // the records this row starts from: black (reset(f1), reset(f2), fwd2c.h:385-386), or the left boundary corner (m+1, b.left)
    RS oH = rs_black(), oG = rs_black(), oG2 = rs_black(), oF = rs_black(), oF2 = rs_black();
    if (row_ok && m + 1 < a.right && m + 1 <= m_left_last && m + 1 + P.lw <= b.left) {
        const GLB unsigned *src = colH + (size_t) (m + 1 - a.left) * 4;
        oH.val = *(const GLB double *) src; oH.dir = (int) src[2]; oH.glb = 0;
    }
    // staging: lanes 0-3 move the four dwords of a record of the strip above (or of a boundary chain) per kind
    auto stage_load = [&](int col, bool wantG, unsigned &rh, unsigned &rg, unsigned &rg2) {
        if (lane < 4) {
            const GLB unsigned *s = (col == b.left && vert0) ? colH + (size_t) (m0 - a.left) * 4 : rowHp + (size_t) col * 4;
            rh = s[lane];
            if (wantG) { rg = rowGp[(size_t) col * 4 + lane]; if (NOLL3) rg2 = rowG2p[(size_t) col * 4 + lane]; }
        }
    };
    auto stage_store = [&](int col, bool wantG, unsigned rh, unsigned rg, unsigned rg2) {
        if (lane < 4) {
            stsc[SLOT_H(col) * 4 + lane] = rh;
            if (wantG) { stsc[(3 + (col & 1)) * 4 + lane] = rg; if (NOLL3) stsc[(5 + (col & 1)) * 4 + lane] = rg2; }
        }
    };
    int avail = prog_up ? 0 : 0x7fffffff;
    const int penc = (pgen & 0x7FF) << 20;
    auto need = [&](const int col) {                       // wave-uniform: every lane polls, nobody branches alone
        const int want = penc | (col < 0xFFFFF ? col : 0xFFFFF);
        if (prog_up && want > avail) {
            avail = g2g_wait_ge(prog_up, want, dbg, failp, ti);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto publish = [&](const int col) {
        if (prog_self) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G2G_POST(prog_self, penc | (col < 0 ? 0 : col < 0xFFFFF ? col : 0xFFFFF));
        }
    };
    if (lane < 28) stsc[lane] = 0;
    team_sync();
    need(cbase + 1 <= c1 ? cbase + 1 : cbase);
    {
        unsigned rh = 0, rg = 0, rg2 = 0;
        stage_load(cbase, false, rh, rg, rg2);
        stage_store(cbase, false, rh, rg, rg2);
        if (cbase + 1 <= c1) {
            stage_load(cbase + 1, vert0, rh, rg, rg2);
            stage_store(cbase + 1, vert0, rh, rg, rg2);
        }
    }
    const double a_cfq = row_ok ? thk_at(a, m)[0] : 0, a_efq = row_ok ? thk_at(a, m)[2] : 0;
    const double pua_row = row_ok ? unpa(P, m, nlo) : 0;                 // fwd2c.h:380 (402 when a.inex.nils)
    SimBlk SB; SB.buf = (GLBV3 double *) simscr; SB.cbase = cbase;      // strip-local column scores
    simblk_fill(P, SB, 0, m0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double sim_cur = 0, bc_cur = 0, be_cur = 0;
    bool have = false;
    RS hu = rs_black(), gu = rs_black(), g2u = rs_black(), hd;
    const bool do_vert = m > a.left;
    const bool wr_rows = mend < a.right;                   // a strip below will read this strip's last row
    team_sync();
    unsigned st_h = 0, st_g = 0, st_g2 = 0;
    bool st_prev = false;
    bool p_act = false; int p_trb = 0; size_t p_tri = 0;
    const int ull = __builtin_amdgcn_readfirstlane(llast);
    int lhi = m0 + llast + P.up + 1; if (lhi > b.right) lhi = b.right;
    int llo = m0 + llast + P.lw; if (llo < b.left) llo = b.left;
    auto flush_rows = [&](const int nl) {                  // the last row's newest corner -> HBM, one dword per lane (4 lanes)
        if (nl >= llo && nl < lhi) {
            const int col = nl + 1;
#pragma unroll
            for (int x = 0; x < (NOLL3 ? 3 : 2); ++x) {
                const RS &r = (x == 0) ? oH : (x == 1) ? oG : oG2;
                const unsigned v0 = (unsigned) __builtin_amdgcn_readlane(__double2loint(r.val), ull);
                const unsigned v1 = (unsigned) __builtin_amdgcn_readlane(__double2hiint(r.val), ull);
                const unsigned v2 = (unsigned) __builtin_amdgcn_readlane(r.dir, ull);
                const unsigned v = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : 0;
                GLB unsigned *dst = (x == 0) ? rowHc : (x == 1) ? rowGc : rowG2c;
                if (lane < 4) dst[(size_t) col * 4 + lane] = v;
            }
        }
    };
    for (int s = 0; s < nsteps; ++s) {
        const int n = cbase + s - lane;
        const int n0 = cbase + s;
        const bool active = row_ok && n >= lo && n < hi;
        if (st_prev) stage_store(n0 + 1, vert0, st_h, st_g, st_g2);
        if (p_act) trace[p_tri] = (uint8_t) p_trb;
        if (wr_rows && s > 0) flush_rows(n0 - 1 - llast);
        if (prog_self && s > 0 && (s & (pint - 1)) == 0) publish(n0 - llast);
        if ((s & 63) == 0) { simblk_fill(P, SB, (s >> 6) + 1, m0, lane); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        hd = hu;
        hu = rs_up(oH); gu = rs_up(oG);
        if (NOLL3) g2u = rs_up(oG2);
        {
            const lu32 *q = stsc + SLOT_H(n0) * 4;
            RS t; t.val = *(const lf64 *) q; t.dir = (int) q[2]; t.glb = 0;
            hd = rs_sel(lane == 0, t, hd);
            q = stsc + SLOT_H(n0 + 1) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2];
            hu = rs_sel(lane == 0, t, hu);
            q = stsc + (3 + ((n0 + 1) & 1)) * 4;
            t.val = *(const lf64 *) q; t.dir = (int) q[2];
            gu = rs_sel(lane == 0, t, gu);
            if (NOLL3) {
                q = stsc + (5 + ((n0 + 1) & 1)) * 4;
                t.val = *(const lf64 *) q; t.dir = (int) q[2];
                g2u = rs_sel(lane == 0, t, g2u);
            }
        }
        double sim_nx = 0, bc_nx = 0, be_nx = 0;
        if (active) {
            if (!have) { sim_cur = *simblk_at(SB, lane, n); bc_cur = bthk[(size_t) (n + 1) * 3]; be_cur = bthk[(size_t) (n + 1) * 3 + 2]; }
            if (n + 1 < hi) { sim_nx = *simblk_at(SB, lane, n + 1); bc_nx = bthk[(size_t) (n + 2) * 3]; be_nx = bthk[(size_t) (n + 2) * 3 + 2]; }
        }
        st_prev = n0 + 1 < hi0 && n0 + 2 <= c1;
        if (st_prev) { need(n0 + 2); stage_load(n0 + 2, vert0, st_h, st_g, st_g2); }
        RS myH = oH, myG = oG, myG2 = oG2;
        if (active) {
            const bool do_hori = n > b.left;
            const bool up_in = do_vert && (n - (m - 1) <= P.up);
            const bool left_in = (n - 1 - m >= P.lw);
            const RS bk = rs_black();
            const RS s_hu = rs_sel(up_in, hu, bk), s_gu = rs_sel(up_in, gu, bk), s_g2u = rs_sel(up_in, g2u, bk);
            const RS s_hl = rs_sel(left_in, oH, bk), s_fl = rs_sel(left_in, oF, bk), s_f2l = rs_sel(left_in, oF2, bk);
            const double pua = a.nils ? unpa(P, m, n) : pua_row;
            const double pub = bc_cur * a_efq * -P.u;                       // unp1(bsi, asi), maln.h:185-187
            // Fwd2c<DPunit>::gapopen (fwd2c.cc:52-91 without DiThk): vgop(cfq of the gapped side's partner ... ) unless the
            // source already runs in that direction
            const double gv = P.basic_gop * (a_cfq * be_cur), gh = P.basic_gop * (bc_cur * a_efq);
            Costs c;
            c.d0 = 0; c.d1 = 0;
            c.gnpv = isvert(s_gu.dir) ? 0. : gv; c.gopv = isvert(s_hu.dir) ? 0. : gv; c.gnpv2 = isvert(s_g2u.dir) ? 0. : gv;
            c.gnph = ishori(s_fl.dir) ? 0. : gh; c.goph = ishori(s_hl.dir) ? 0. : gh; c.gnph2 = ishori(s_f2l.dir) ? 0. : gh;
            const Dec d = v3_decide<1, NOLL3>(P, c, hd, s_hu, s_gu, s_g2u, s_hl, s_fl, s_f2l, do_vert, do_hori, sim_cur, pua, pub);
            int trb = 0;
            v3_outputs<0, NOLL3>(d, 0, 0, do_vert, do_hori, myH, myG, myG2, oF, oF2, trb);
            const int dd = m + n;
            int mlo, mhi;
            diag_rows(dd, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
            p_tri = (size_t) (dd - P.d0) * P.tstride + (m - mlo);
            p_trb = trb;
            sim_cur = sim_nx; bc_cur = bc_nx; be_cur = be_nx; have = (n + 1 < hi);
            if (m == a.right - 1 && n == b.right - 1) *P.score = myH.val;
        }
        p_act = active;
        oH = myH; oG = myG; oG2 = myG2;
        team_sync();
    }
    if (p_act) trace[p_tri] = (uint8_t) p_trb;
    if (wr_rows) flush_rows(cbase + nsteps - 1 - llast);
    publish(0xFFFFF);
}

#define V7_KERNEL(NAME, N3)                                                                         \
extern "C" __global__ void __launch_bounds__(64)                                                    \
NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int pint, double *simscr) \
{                                                                                                   \
    __shared__ __attribute__((aligned(16))) unsigned v7_lds[64 + 32];                               \
    li32 *s_vals = (li32 *) ((lchar *) v7_lds + 128);                                               \
    for (;;) {                                                                                      \
        s_vals[threadIdx.x] = atomicAdd(qhead, threadIdx.x == 0 ? 1 : 0);                           \
        __syncthreads();                                                                            \
        const int t = __builtin_amdgcn_readfirstlane(s_vals[0]);                                    \
        __syncthreads();                                                                            \
        if (t >= ntiles) break;                                                                     \
        const V2Tile T = tiles[t];                                                                  \
        if (T.ti < 0) {           /* a boundary chain */                                            \
            v7_chain_tile(probs[T.prob], T.ti, done + T.self, gen);                                 \
            __syncthreads();                                                                        \
            continue;                                                                               \
        }                                                                                           \
        int *failp = done + done[G2G_HDR + 2] + T.prob;                                             \
        if (threadIdx.x == 0) s_vals[0] = g2g_dp_failed(failp) ? 1 : 0;                             \
        __syncthreads();                                                                            \
        const int dp_dead = s_vals[0];                                                              \
        __syncthreads();                                                                            \
        if (dp_dead) {                                                                              \
            if (threadIdx.x == 0) G2G_POST(done + T.self, ((gen & 0x7FF) << 20) | 0xFFFFF); \
            __syncthreads();                                                                        \
            continue;                                                                               \
        }                                                                                           \
        const int *pl = T.dep_left >= 0 ? done + T.dep_left : (const int *) 0;                      \
        const int *pu = T.dep_up >= 0 ? done + T.dep_up : (const int *) 0;                          \
        v7_strip<N3>(probs[T.prob], (lchar *) v7_lds, T.ti, T.nsteps, pu, done + T.self, done + G2G_HDR, gen, pint, pl, \
                     simscr + (size_t) blockIdx.x * G2G_SIMBLK_STRIDE, failp);                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __syncthreads();                                                                            \
    }                                                                                               \
}
#ifdef G2G_TU_V78
V7_KERNEL(g2g_v7_ngp2, false)
V7_KERNEL(g2g_v7_ngp3, true)
#else
extern "C" __global__ void g2g_v7_ngp2(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int pint, double *simscr); extern "C" __global__ void g2g_v7_ngp3(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int pint, double *simscr);
#endif
