// g2g_kernels_v8.hip -- strip kernel of the NAIVE record type: Fwd2c<DPunit_nv> (alignment mode NTV_ALB: so few members
// that the reference keeps one running gap length per MEMBER instead of gap profiles), Noll 2/3.  Same recurrence and
// arithmetic order as g2g_forward_kernel<3> (reference src/fwd2c.h:359-482; gapopen of _nv src/fwd2c.cc:107-111 = PwdM::crg??,
// src/maln2.cc:881-1024 and :1454-1614; update = elongap, src/mgaps.cc:442-451).
//
// selAlnMode (maln2.cc:81-154) picks the mode only while 2 x (smaller group) + (larger group) < 8, i.e. for member counts
// (1,1) ... (5,1), (2,2), (3,2): at most six gap lengths per record.  They live in six VGPRs -- a's members in g[0], g[1],
// ..., b's members from the top, g[5], g[4], ... -- so that every index is a compile-time constant after unrolling; the
// member counts only switch (wave-uniform) guards.  The gap-open costs of the three records a direction reads (H, G, G2
// above; H, F, F2 to the left) are evaluated by ONE pass over the member pairs.
//
// Mapping = g2g_kernels_v7.hip: a strip of 64 rows belongs to one wave, lane t owns row m0 + t one column behind lane
// t-1, records move down the lanes by DPP, the strip's last row goes to HBM (48-byte records: value, direction, six
// lengths) for the strip below, which follows on a progress counter; strips make their own column scores (SimBlk).
// v1 ran these DPs with one workgroup per DP and every record and every length in HBM.
#include <hip/hip_runtime.h>

#define V8_REC 12                                            // dwords of a record in HBM / in the staging slots
struct NL { int g[6]; };
__device__ __forceinline__ NL nl_zero() { NL r; _Pragma("unroll") for (int k = 0; k < 6; ++k) r.g[k] = 0; return r; }
__device__ __forceinline__ NL nl_up(const NL &x) { NL r; _Pragma("unroll") for (int k = 0; k < 6; ++k) r.g[k] = dpp_up1(x.g[k]); return r; }
__device__ __forceinline__ NL nl_sel(const bool c, const NL &x, const NL &y) { NL r; _Pragma("unroll") for (int k = 0; k < 6; ++k) r.g[k] = c ? x.g[k] : y.g[k]; return r; }
#define V8_GA(L, i) ((L).g[i])
#define V8_GB(L, j) ((L).g[5 - (j)])

// what gapopen / update read of one position of a group: gap density and post-gap density per member (seq.h gapdensity /
// postgapdensity, precomputed by the host), bit i of ng = member i holds a residue there (code > 1)
struct NPos { double gd[5], pg[5]; unsigned ng; bool two; };
__device__ __forceinline__ void v8_load_pos(NPos &X, const DevSide &s, const int pos, const int many)
{
    const GLB double *gd = glb(s.gapdens + (size_t) (pos + 1) * many), *pg = glb(s.postgapdens + (size_t) (pos + 1) * many);
    const GLB uint8_t *rs = glb(res_at(s, pos));
    X.ng = 0;
    _Pragma("unroll") for (int k = 0; k < 5; ++k) {
        X.gd[k] = 0; X.pg[k] = 0;
        if (k < many) { X.gd[k] = gd[k]; X.pg[k] = pg[k]; X.ng |= (rs[k] > 1 ? 1u : 0u) << k; }
    }
    X.two = rs[0] != 0 && rs[1] != 0;                          // crg21w's `*bs && bs[1]` (maln2.cc:1518; with one member bs[1] is the next column)
}

// elongap (mgaps.cc:442-451) on both sides: a member that shows a residue restarts at 0, everybody else grows
template <int D3>
__device__ __forceinline__ NL v8_update(const NL &s, const unsigned ang, const unsigned bng, const int an, const int bn)
{
    NL d = s;
    _Pragma("unroll") for (int i = 0; i < 5; ++i)
        if (i < an) V8_GA(d, i) = (D3 >= 0 && ((ang >> i) & 1)) ? 0 : V8_GA(s, i) + 1;
    _Pragma("unroll") for (int j = 0; j < 5; ++j)
        if (j < bn) V8_GB(d, j) = (D3 <= 0 && ((bng >> j) & 1)) ? 0 : V8_GB(s, j) + 1;
    return d;
}

// PwdM::crg?? for NR records at once (out[r]); wa / wb: the members' weights, 1.0 for the unweighted variants (x * 1.0 == x)
#define V8_ADD(acc, cond, term) acc = (cond) ? acc + (term) : acc
template <int D3, int NR>
__device__ __forceinline__ void v8_crg(const int kind, const double gop, const int an, const int bn, const NPos &A, const NPos &B,
                                       const double *wa, const double *wb, const NL &r0, const NL &r1, const NL &r2, double *out)
{
#define V8_R(r) ((r) == 0 ? r0 : (r) == 1 ? r1 : r2)
    double g[NR];
    _Pragma("unroll") for (int r = 0; r < NR; ++r) g[r] = 0;
#define NGA(i) ((A.ng >> (i)) & 1)
#define NGB(j) ((B.ng >> (j)) & 1)
    if (kind == 11) {                                         // crg11, maln2.cc:881-902
        _Pragma("unroll") for (int r = 0; r < NR; ++r) {
            const int ga = V8_GA(V8_R(r), 0), gb = V8_GB(V8_R(r), 0);
            if (D3 == 0) out[r] = (NGA(0) && B.gd[0] > 0 && ga >= gb) ? B.gd[0] * gop : (NGB(0) && A.gd[0] > 0 && gb >= ga) ? A.gd[0] * gop : 0;
            else if (D3 > 0) out[r] = (B.pg[0] > 0 && ga >= gb) ? B.pg[0] * gop : 0;
            else out[r] = (A.pg[0] > 0 && gb >= ga) ? A.pg[0] * gop : 0;
        }
        return;
    }
    if (kind == 120 || kind == 121) {                         // crg12i :904-939, crg12w :1454-1490 (a has one member)
        _Pragma("unroll") for (int j = 0; j < 5; ++j) if (j < bn) {
            _Pragma("unroll") for (int r = 0; r < NR; ++r) {
                const int ga = V8_GA(V8_R(r), 0), gb = V8_GB(V8_R(r), j);
                if (D3 == 0) {
                    V8_ADD(g[r], NGA(0) && B.gd[j] > 0 && ga >= gb, wb[j] * B.gd[j]);
                    V8_ADD(g[r], !NGA(0) && A.gd[0] > 0 && NGB(j) && gb >= ga, wb[j] * A.gd[0]);
                } else if (D3 > 0) V8_ADD(g[r], NGA(0) && B.pg[j] > 0 && ga >= gb, wb[j] * B.pg[j]);
                else V8_ADD(g[r], A.pg[0] > 0 && NGB(j) && gb >= ga, wb[j] * A.pg[0]);
            }
        }
    } else if (kind == 210 || kind == 211) {                  // crg21i :941-976, crg21w :1492-1528 (b has one member)
        const bool w = kind & 1;
        _Pragma("unroll") for (int i = 0; i < 5; ++i) if (i < an) {
            _Pragma("unroll") for (int r = 0; r < NR; ++r) {
                const int ga = V8_GA(V8_R(r), i), gb = V8_GB(V8_R(r), 0);
                if (D3 == 0) {
                    V8_ADD(g[r], NGB(0) && A.gd[i] > 0 && gb >= ga, wa[i] * A.gd[i]);
                    V8_ADD(g[r], !NGB(0) && B.gd[0] > 0 && NGA(i) && ga >= gb, wa[i] * B.gd[0]);
                } else if (D3 < 0) V8_ADD(g[r], NGB(0) && A.pg[i] > 0 && gb >= ga, wa[i] * A.pg[i]);
                else V8_ADD(g[r], (!w || B.two) && B.pg[0] > 0 && NGA(i) && ga >= gb, wa[i] * B.pg[0]);
            }
        }
    } else if (kind == 220) {                                 // crg22i :978-1024: one flat sum over the member pairs
        if (D3 >= 0) {
            _Pragma("unroll") for (int i = 0; i < 3; ++i) if (i < an)
                _Pragma("unroll") for (int j = 0; j < 3; ++j) if (j < bn)
                    _Pragma("unroll") for (int r = 0; r < NR; ++r) {
                        const int ga = V8_GA(V8_R(r), i), gb = V8_GB(V8_R(r), j);
                        if (D3 == 0) {
                            V8_ADD(g[r], NGA(i) && B.gd[j] > 0 && ga >= gb, B.gd[j]);
                            V8_ADD(g[r], !NGA(i) && A.gd[i] > 0 && NGB(j) && gb >= ga, A.gd[i]);
                        } else V8_ADD(g[r], NGA(i) && B.pg[j] > 0 && ga >= gb, B.pg[j]);
                    }
        } else {
            _Pragma("unroll") for (int j = 0; j < 3; ++j) if (j < bn)
                _Pragma("unroll") for (int i = 0; i < 3; ++i) if (i < an)
                    _Pragma("unroll") for (int r = 0; r < NR; ++r)
                        V8_ADD(g[r], NGB(j) && A.pg[i] > 0 && V8_GB(V8_R(r), j) >= V8_GA(V8_R(r), i), A.pg[i]);
        }
    } else {                                                  // crg22w :1551-1614: inner sums, weighted by the outer member
        if (D3 >= 0) {
            _Pragma("unroll") for (int i = 0; i < 3; ++i) if (i < an) {
                double s[NR];
                _Pragma("unroll") for (int r = 0; r < NR; ++r) s[r] = 0;
                _Pragma("unroll") for (int j = 0; j < 3; ++j) if (j < bn)
                    _Pragma("unroll") for (int r = 0; r < NR; ++r) {
                        const int ga = V8_GA(V8_R(r), i), gb = V8_GB(V8_R(r), j);
                        if (D3 == 0) {
                            V8_ADD(s[r], NGA(i) && B.gd[j] > 0 && ga >= gb, wb[j] * B.gd[j]);
                            V8_ADD(s[r], !NGA(i) && A.gd[i] > 0 && NGB(j) && gb >= ga, wb[j] * A.gd[i]);
                        } else V8_ADD(s[r], B.pg[j] > 0 && ga >= gb, wb[j] * B.pg[j]);
                    }
                // D3 == 0: every member adds its (possibly empty) sum; D3 > 0: only members that hold a residue (:1590-1598)
                _Pragma("unroll") for (int r = 0; r < NR; ++r) V8_ADD(g[r], D3 == 0 || NGA(i), s[r] * wa[i]);
            }
        } else {
            _Pragma("unroll") for (int j = 0; j < 3; ++j) if (j < bn) {
                double s[NR];
                _Pragma("unroll") for (int r = 0; r < NR; ++r) s[r] = 0;
                _Pragma("unroll") for (int i = 0; i < 3; ++i) if (i < an)
                    _Pragma("unroll") for (int r = 0; r < NR; ++r)
                        V8_ADD(s[r], A.pg[i] > 0 && V8_GB(V8_R(r), j) >= V8_GA(V8_R(r), i), wa[i] * A.pg[i]);
                _Pragma("unroll") for (int r = 0; r < NR; ++r) V8_ADD(g[r], NGB(j), s[r] * wb[j]);
            }
        }
    }
#undef NGA
#undef NGB
#undef V8_R
    _Pragma("unroll") for (int r = 0; r < NR; ++r) out[r] = g[r] * gop;
}

// a record of the row / column buffers: {value, direction, -, g[0..5], -, -}
__device__ __forceinline__ void v8_rec_store(GLB unsigned *dst, const RS &r, const NL &l)
{
    typedef unsigned v8_u4 __attribute__((ext_vector_type(4)));
    v8_u4 q0, q1, q2;
    q0.x = (unsigned) __double2loint(r.val); q0.y = (unsigned) __double2hiint(r.val); q0.z = (unsigned) r.dir; q0.w = 0;
    q1.x = l.g[0]; q1.y = l.g[1]; q1.z = l.g[2]; q1.w = l.g[3];
    q2.x = l.g[4]; q2.y = l.g[5]; q2.z = 0; q2.w = 0;
    GLB v8_u4 *d = (GLB v8_u4 *) dst;
    d[0] = q0; d[1] = q1; d[2] = q2;
}

// the two boundary chains of Fwd2c::initB (src/fwd2c.h:138-176) for DPunit_nv: one lane walks
__device__ __forceinline__ void v8_chain_tile(const DevProb &Pmem, const int which, int *prog, const int pgen)
{
    if (threadIdx.x != 0) return;
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int an = a.many, bn = b.many, kind = P.crg2_kind;
    const int penc = (pgen & 0x7FF) << 20;
    double wa[5], wb[5];
    _Pragma("unroll") for (int k = 0; k < 5; ++k) { wa[k] = ((kind & 1) && k < an) ? a.weight[k] : 1.; wb[k] = ((kind & 1) && k < bn) ? b.weight[k] : 1.; }
    RS r; r.val = 0; r.dir = D_DIAG; r.glb = 0;
    NL l = nl_zero();
    NPos A, B;
    if (which == -1) {                                     // top row: corners (a.left, n), n = b.left .. a.left + rr
        GLB unsigned *rowH = glbw((unsigned *) P.v2_rowH + 2 * (size_t) P.v2_rowstride * V8_REC);
        int rrt = b.right - a.left; if (P.up < rrt) rrt = P.up;
        const int nlast = a.left + rrt, ai = a.left - 1;
        v8_rec_store(rowH + (size_t) b.left * V8_REC, r, l);
        v8_load_pos(A, a, ai, an);
        for (int n = b.left + 1; n <= nlast; ++n) {
            const int bi = n - 1;
            v8_load_pos(B, b, bi, bn);
            const double pub = unpb(P, bi, ai);
            double gnp;
            v8_crg<-1, 1>(kind, P.basic_gop, an, bn, A, B, wa, wb, l, l, l, &gnp);       // gapopen(prv, -1)
            gnp = (n - b.left < P.codonk1) ? gnp + pub : (P.v2divv1 * gnp + P.u2divu1 * pub);
            r.dir = isvert(r.dir) ? D_NEWH : D_HORI;
            r.val = r.val + gnp;
            l = v8_update<-1>(l, A.ng, B.ng, an, bn);
            v8_rec_store(rowH + (size_t) n * V8_REC, r, l);
            if (((n - b.left) & 63) == 0) chain_publish(prog, penc, n);
        }
    } else {                                               // left column: corners (m, b.left), m = a.left .. b.left - rr
        GLB unsigned *colH = glbw((unsigned *) P.v2_colH);
        int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;
        const int mlast = b.left - rrl, bi = b.left - 1;
        v8_load_pos(B, b, bi, bn);
        for (int m = a.left + 1; m <= mlast; ++m) {
            const int ai = m - 1;
            v8_load_pos(A, a, ai, an);
            const double pua = unpa(P, ai, bi);
            double gnp;
            v8_crg<1, 1>(kind, P.basic_gop, an, bn, A, B, wa, wb, l, l, l, &gnp);        // gapopen(prv, 1)
            gnp = (m - a.left < P.codonk1) ? gnp + pua : (P.v2divv1 * gnp + P.u2divu1 * pua);
            r.dir = ishori(r.dir) ? D_NEWV : D_VERT;
            r.val = r.val + gnp;
            l = v8_update<1>(l, A.ng, B.ng, an, bn);
            v8_rec_store(colH + (size_t) (m - a.left) * V8_REC, r, l);
            if (((m - a.left) & 63) == 0) chain_publish(prog, penc, m - a.left);
        }
    }
    chain_publish(prog, penc, 0xFFFFF);
}

#define V8_SLOT(kind_, col) ((kind_) == 0 ? SLOT_H(col) : (kind_) == 1 ? 3 + ((col) & 1) : 5 + ((col) & 1))
template <bool NOLL3>
__device__ __forceinline__ void v8_strip(const DevProb &Pmem, lchar *lds, const int ti, const int nsteps,
                                         const int *prog_up, int *prog_self, int *dbg, const int pgen, const int pint, const int *prog_left,
                                         double *simscr, int *failp)
{
    DevProb P;
    uni_prob(P, Pmem);
    const DevSide &a = P.a, &b = P.b;
    const int an = a.many, bn = b.many, kind = P.crg2_kind;
    const int lane = threadIdx.x;                          // blockDim.x == 64
    lu32 *const stsc = (lu32 *) lds;                       // staging records of lane 0's upper neighbours: H ring 0-2, G 3-4, G2 5-6
    const size_t rbuf = (size_t) P.v2_rowstride * V8_REC;
    const int bprev = (ti + 2) % 3, bcur = ti % 3;
    const GLB unsigned *rowHp = glb((const unsigned *) P.v2_rowH + bprev * rbuf), *rowGp = glb((const unsigned *) P.v2_rowG + bprev * rbuf);
    const GLB unsigned *rowG2p = NOLL3 ? glb((const unsigned *) P.v2_rowG2 + bprev * rbuf) : 0;
    GLB unsigned *rowHc = glbw((unsigned *) P.v2_rowH + bcur * rbuf), *rowGc = glbw((unsigned *) P.v2_rowG + bcur * rbuf);
    GLB unsigned *rowG2c = NOLL3 ? glbw((unsigned *) P.v2_rowG2 + bcur * rbuf) : 0;
    const GLB unsigned *colH = glb((const unsigned *) P.v2_colH);
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
    NL lH = nl_zero(), lG = nl_zero(), lG2 = nl_zero(), lF = nl_zero(), lF2 = nl_zero();
    if (row_ok && m + 1 < a.right && m + 1 <= m_left_last && m + 1 + P.lw <= b.left) {
        const GLB unsigned *src = colH + (size_t) (m + 1 - a.left) * V8_REC;
        oH.val = *(const GLB double *) src; oH.dir = (int) src[2]; oH.glb = 0;
        _Pragma("unroll") for (int k = 0; k < 6; ++k) lH.g[k] = (int) src[4 + k];
    }
    // staging: lanes 0-11 move the dwords of a record of the strip above (or of a boundary chain) per kind
    auto stage_load = [&](int col, bool wantG, unsigned &rh, unsigned &rg, unsigned &rg2) {
        if (lane < V8_REC) {
            const GLB unsigned *s = (col == b.left && vert0) ? colH + (size_t) (m0 - a.left) * V8_REC : rowHp + (size_t) col * V8_REC;
            rh = s[lane];
            if (wantG) { rg = rowGp[(size_t) col * V8_REC + lane]; if (NOLL3) rg2 = rowG2p[(size_t) col * V8_REC + lane]; }
        }
    };
    auto stage_store = [&](int col, bool wantG, unsigned rh, unsigned rg, unsigned rg2) {
        if (lane < V8_REC) {
            stsc[V8_SLOT(0, col) * V8_REC + lane] = rh;
            if (wantG) { stsc[V8_SLOT(1, col) * V8_REC + lane] = rg; if (NOLL3) stsc[V8_SLOT(2, col) * V8_REC + lane] = rg2; }
        }
    };
    auto stage_get = [&](const int slot, RS &r, NL &l) {
        const lu32 *q = stsc + slot * V8_REC;
        r.val = *(const lf64 *) q; r.dir = (int) q[2]; r.glb = 0;
        _Pragma("unroll") for (int k = 0; k < 6; ++k) l.g[k] = (int) q[4 + k];
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
    for (int k = lane; k < 7 * V8_REC; k += 64) stsc[k] = 0;
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
    double wa[5], wb[5];                                   // wave-uniform: scalar registers
    _Pragma("unroll") for (int k = 0; k < 5; ++k) { wa[k] = ((kind & 1) && k < an) ? a.weight[k] : 1.; wb[k] = ((kind & 1) && k < bn) ? b.weight[k] : 1.; }
    NPos A;
    v8_load_pos(A, a, row_ok ? m : a.right - 1, an);
    const double a_efq = row_ok ? thk_at(a, m)[2] : 0;
    const double pua_row = row_ok ? unpa(P, m, nlo) : 0;                 // fwd2c.h:380 (402 when a.inex.nils)
    SimBlk SB; SB.buf = (GLBV3 double *) simscr; SB.cbase = cbase;      // strip-local column scores
    simblk_fill(P, SB, 0, m0, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const GLB double *bthk = glb(b.thk);
    double sim_cur = 0, bc_cur = 0;
    NPos B, Bnx;
    bool have = false;
    RS hu = rs_black(), gu = rs_black(), g2u = rs_black(), hd;
    NL lhu = nl_zero(), lgu = nl_zero(), lg2u = nl_zero(), lhd;
    const bool do_vert = m > a.left;
    const bool wr_rows = mend < a.right;                   // a strip below will read this strip's last row
    team_sync();
    unsigned st_h = 0, st_g = 0, st_g2 = 0;
    bool st_prev = false;
    bool p_act = false; int p_trb = 0; size_t p_tri = 0;
    int lhi = m0 + llast + P.up + 1; if (lhi > b.right) lhi = b.right;
    int llo = m0 + llast + P.lw; if (llo < b.left) llo = b.left;
    auto flush_rows = [&](const int nl) {                  // the last row's newest corner -> HBM, by the lane that owns it
        if (lane == llast && nl >= llo && nl < lhi) {
            const size_t o = (size_t) (nl + 1) * V8_REC;
            v8_rec_store(rowHc + o, oH, lH);
            v8_rec_store(rowGc + o, oG, lG);
            if (NOLL3) v8_rec_store(rowG2c + o, oG2, lG2);
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
        hd = hu; lhd = lhu;
        hu = rs_up(oH); gu = rs_up(oG); lhu = nl_up(lH); lgu = nl_up(lG);
        if (NOLL3) { g2u = rs_up(oG2); lg2u = nl_up(lG2); }
        {
            RS t; NL tl;
            stage_get(V8_SLOT(0, n0), t, tl);
            hd = rs_sel(lane == 0, t, hd); lhd = nl_sel(lane == 0, tl, lhd);
            stage_get(V8_SLOT(0, n0 + 1), t, tl);
            hu = rs_sel(lane == 0, t, hu); lhu = nl_sel(lane == 0, tl, lhu);
            stage_get(V8_SLOT(1, n0 + 1), t, tl);
            gu = rs_sel(lane == 0, t, gu); lgu = nl_sel(lane == 0, tl, lgu);
            if (NOLL3) {
                stage_get(V8_SLOT(2, n0 + 1), t, tl);
                g2u = rs_sel(lane == 0, t, g2u); lg2u = nl_sel(lane == 0, tl, lg2u);
            }
        }
        double sim_nx = 0, bc_nx = 0;
        if (active) {
            if (!have) { sim_cur = *simblk_at(SB, lane, n); bc_cur = bthk[(size_t) (n + 1) * 3]; v8_load_pos(B, b, n, bn); }
            if (n + 1 < hi) { sim_nx = *simblk_at(SB, lane, n + 1); bc_nx = bthk[(size_t) (n + 2) * 3]; v8_load_pos(Bnx, b, n + 1, bn); }
        }
        st_prev = n0 + 1 < hi0 && n0 + 2 <= c1;
        if (st_prev) { need(n0 + 2); stage_load(n0 + 2, vert0, st_h, st_g, st_g2); }
        RS myH = oH, myG = oG, myG2 = oG2;
        NL nH = lH, nG = lG, nG2 = lG2;
        if (active) {
            const bool do_hori = n > b.left;
            const bool up_in = do_vert && (n - (m - 1) <= P.up);
            const bool left_in = (n - 1 - m >= P.lw);
            const RS bk = rs_black();
            const NL zl = nl_zero();
            const RS s_hu = rs_sel(up_in, hu, bk), s_gu = rs_sel(up_in, gu, bk), s_g2u = rs_sel(up_in, g2u, bk);
            const RS s_hl = rs_sel(left_in, oH, bk), s_fl = rs_sel(left_in, oF, bk), s_f2l = rs_sel(left_in, oF2, bk);
            const NL l_hu = nl_sel(up_in, lhu, zl), l_gu = nl_sel(up_in, lgu, zl), l_g2u = nl_sel(up_in, lg2u, zl);
            const NL l_hl = nl_sel(left_in, lH, zl), l_fl = nl_sel(left_in, lF, zl), l_f2l = nl_sel(left_in, lF2, zl);
            const double pua = a.nils ? unpa(P, m, n) : pua_row;
            const double pub = bc_cur * a_efq * -P.u;                       // unp1(bsi, asi), maln.h:185-187
            Costs c;
            c.d1 = 0;
            double cd[1], cv[3], ch[3];
            v8_crg<0, 1>(kind, P.basic_gop, an, bn, A, B, wa, wb, lhd, lhd, lhd, cd);
            v8_crg<1, 3>(kind, P.basic_gop, an, bn, A, B, wa, wb, l_gu, l_hu, l_g2u, cv);
            v8_crg<-1, 3>(kind, P.basic_gop, an, bn, A, B, wa, wb, l_fl, l_hl, l_f2l, ch);
            c.d0 = cd[0];
            c.gnpv = cv[0]; c.gopv = cv[1]; c.gnpv2 = cv[2];
            c.gnph = ch[0]; c.goph = ch[1]; c.gnph2 = ch[2];
            const Dec d = v3_decide<1, NOLL3>(P, c, hd, s_hu, s_gu, s_g2u, s_hl, s_fl, s_f2l, do_vert, do_hori, sim_cur, pua, pub);
            int trb = 0;
            v3_outputs<0, NOLL3>(d, 0, 0, do_vert, do_hori, myH, myG, myG2, oF, oF2, trb);
            // the lengths follow their records (update: fwd2c.cc:114-128)
            const NL uD = v8_update<0>(lhd, A.ng, B.ng, an, bn);
            const NL uG = v8_update<1>(nl_sel(d.g_from_h, l_hu, l_gu), A.ng, B.ng, an, bn);
            const NL uF = v8_update<-1>(nl_sel(d.f_from_h, l_hl, l_fl), A.ng, B.ng, an, bn);
            NL uG2 = zl, uF2 = zl;
            if (NOLL3) {
                uG2 = v8_update<1>(nl_sel(d.g2_from_h, l_hu, l_g2u), A.ng, B.ng, an, bn);
                uF2 = v8_update<-1>(nl_sel(d.f2_from_h, l_hl, l_f2l), A.ng, B.ng, an, bn);
            }
            if (do_vert) nG = uG;
            if (NOLL3 && do_vert) nG2 = uG2;
            if (do_hori) lF = uF;
            if (NOLL3 && do_hori) lF2 = uF2;
            nH = nl_sel(d.win == 0, uD, nl_sel(d.win == 1, uG, nl_sel(d.win == 3, uF, NOLL3 ? nl_sel(d.win == 2, uG2, uF2) : uF)));
            const int dd = m + n;
            int mlo, mhi;
            diag_rows(dd, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
            p_tri = (size_t) (dd - P.d0) * P.tstride + (m - mlo);
            p_trb = trb;
            sim_cur = sim_nx; bc_cur = bc_nx; B = Bnx; have = (n + 1 < hi);
            if (m == a.right - 1 && n == b.right - 1) *P.score = myH.val;
        }
        p_act = active;
        oH = myH; oG = myG; oG2 = myG2;
        lH = nH; lG = nG; lG2 = nG2;
        team_sync();
    }
    if (p_act) trace[p_tri] = (uint8_t) p_trb;
    if (wr_rows) flush_rows(cbase + nsteps - 1 - llast);
    publish(0xFFFFF);
}

#define V8_KERNEL(NAME, N3)                                                                         \
extern "C" __global__ void __launch_bounds__(64)                                                    \
NAME(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int pint, double *simscr) \
{                                                                                                   \
    __shared__ __attribute__((aligned(16))) unsigned v8_lds[96 + 64];                               \
    li32 *s_vals = (li32 *) ((lchar *) v8_lds + 384);                                               \
    for (;;) {                                                                                      \
        s_vals[threadIdx.x] = atomicAdd(qhead, threadIdx.x == 0 ? 1 : 0);                           \
        __syncthreads();                                                                            \
        const int t = __builtin_amdgcn_readfirstlane(s_vals[0]);                                    \
        __syncthreads();                                                                            \
        if (t >= ntiles) break;                                                                     \
        const V2Tile T = tiles[t];                                                                  \
        if (T.ti < 0) {           /* a boundary chain */                                            \
            v8_chain_tile(probs[T.prob], T.ti, done + T.self, gen);                                 \
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
        v8_strip<N3>(probs[T.prob], (lchar *) v8_lds, T.ti, T.nsteps, pu, done + T.self, done + G2G_HDR, gen, pint, pl, \
                     simscr + (size_t) blockIdx.x * G2G_SIMBLK_STRIDE, failp);                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        __syncthreads();                                                                            \
    }                                                                                               \
}
#ifdef G2G_TU_V78
V8_KERNEL(g2g_v8_ntv2, false)
V8_KERNEL(g2g_v8_ntv3, true)
#else
extern "C" __global__ void g2g_v8_ntv2(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int pint, double *simscr); extern "C" __global__ void g2g_v8_ntv3(const DevProb *probs, const V2Tile *tiles, int ntiles, int *qhead, int *done, int gen, int pint, double *simscr);
#endif
