// g2g_kernels.hip -- gfx950 kernels of the group-to-group DP engine.
//
// What they compute: Fwd2c<recd_t>::forwardB (reference src/fwd2c.h:359-482, initB :138-176) with the
// record-type specialisations of src/fwd2c.cc (DPunit :52-102, _nv :107-147, _hf :152-198, _pf :203-251)
// and the gap-state algebra of src/gfreq.cc:493-605 -- but scheduled as an ANTI-DIAGONAL wavefront:
// all cells with the same m + n are independent (each reads the corners (m,n), (m,n+1), (m+1,n) written
// on the two previous anti-diagonals), one workgroup sweeps one DP, one barrier per anti-diagonal.
// The reference's diagonal-indexed row buffer maps onto this unchanged: cell (m,n) owns index r = n - m
// of H/G/G2/F/F2, reads r-1 and r+1 (other parity class), so no double buffering is needed.
// The Vmf linked list (src/vmf.h) is replaced by one trace byte per cell + a backtrack kernel that
// re-derives exactly the records Vmf::traceback(-1) would return.
//
// Arithmetic: IEEE doubles, reference operand order, built with -ffp-contract=off (no FMA), so scores
// are bit-identical to the reference CPU path.  No MFMA: the recurrence is scalar.
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include "g2g_device.h"

#define NEVSEL (-(DBL_MAX / 16 * 7))      /* reference src/cmn.h:57 */
#ifndef G2G_FWD_THREADS
#define G2G_FWD_THREADS 256
#endif
#ifndef G2G_V2_THREADS
#define G2G_V2_THREADS 256
#endif
#ifndef G2G_V2_MINWAVES
#define G2G_V2_MINWAVES 4
#endif
#ifndef G2G_V2_TILE_COLS
#define G2G_V2_TILE_COLS 256
#endif

// TraceBackDir values Fwd2c produces (src/aln.h:47-52) and their classes (:59-61)
enum { D_DEAD = 0, D_DIAG = 2, D_NEWD = 3, D_VERT = 4, D_HORI = 8, D_NEWV = 12, D_NEWH = 13 };
__device__ __forceinline__ bool isdiag(int d) { return d == D_DIAG || d == D_NEWD; }
__device__ __forceinline__ bool isvert(int d) { return d == D_VERT || d == D_NEWV; }
__device__ __forceinline__ bool ishori(int d) { return d == D_HORI || d == D_NEWH; }

// trace byte: bits 0-3 final H.dir, bit 4 G extended (else opened from H), bit 5 G2 ext, bit 6 F ext,
// bit 7 F2 ext.  H came from the "2" (long-gap) record iff ... see T_SEL2 (stored in a second nibble
// would not fit, so dir is compressed to 3 bits):
enum { T_DIRMASK = 7, T_SEL2 = 8, T_GEXT = 16, T_G2EXT = 32, T_FEXT = 64, T_F2EXT = 128 };
__device__ __forceinline__ int dir2code(int d)
{   // DIAG 1, NEWD 2, VERT 3, NEWV 4, HORI 5, NEWH 6
    return d == D_DIAG ? 1 : d == D_NEWD ? 2 : d == D_VERT ? 3 : d == D_NEWV ? 4 : d == D_HORI ? 5 : d == D_NEWH ? 6 : 0;
}

// ---- views ------------------------------------------------------------------------------------
struct SList { const int *glen; const double *freq; };            // one static GFREQ list
__device__ __forceinline__ SList gfq_at(const DevSide &s, int view, int pos)
{
    int o = s.off[view][pos + 1];
    SList l; l.glen = s.glen[view] + o; l.freq = s.freq[view] + o;
    return l;
}
struct DList {                                                    // one dynamic IDELTA list
    int2 *p; int s;
    __device__ __forceinline__ int glen(int k) const { return p[(size_t) k * s].x; }
    __device__ __forceinline__ int nins(int k) const { return p[(size_t) k * s].y; }
    __device__ __forceinline__ void set(int k, int g, int n) const { p[(size_t) k * s] = make_int2(g, n); }
};
struct Rec { int x, i; };                                         // buffer X, slot i = r - (lw - 1)

__device__ __forceinline__ DList dla_of(const DevProb &P, Rec r) { DList d; d.p = P.dla[r.x] + r.i; d.s = P.width; return d; }
__device__ __forceinline__ DList dlb_of(const DevProb &P, Rec r) { DList d; d.p = P.dlb[r.x] + r.i; d.s = P.width; return d; }
__device__ __forceinline__ double &val_of(const DevProb &P, Rec r) { return P.val[r.x][r.i]; }
__device__ __forceinline__ int dir_of(const DevProb &P, Rec r) { return P.dir[r.x][r.i]; }

__device__ __forceinline__ const double *thk_at(const DevSide &s, int pos) { return s.thk + (size_t)(pos + 1) * 3; }
__device__ __forceinline__ const uint8_t *res_at(const DevSide &s, int pos) { return s.seq + (size_t)(pos + 1) * s.many; }
__device__ __forceinline__ const double *vss_at(const DevSide &s, int pos) { return s.pseq + (size_t)(pos + 1) * s.nelm; }

// ---- gap-state algebra (reference src/gfreq.cc) -------------------------------------------------
// GapLenSD, gfreq.h:67
__device__ __forceinline__ int gaplen_sd(int g, const DList dl)
{
    int k = 0;
    while (g >= dl.glen(k + 1)) ++k;
    return g + dl.nins(k);
}
// newgap(cf, dlc, df, dld), gfreq.cc:507-521
__device__ double newgap4(const SList cf, const DList dlc, const SList df, const DList dld)
{
    double g = 0;
    int ci = 0;
    for (int di = 0; df.glen[di] >= 0; ++di) {
        int j = gaplen_sd(df.glen[di], dld);
        for ( ; cf.glen[ci] >= 0; ++ci) {
            int i = gaplen_sd(cf.glen[ci], dlc);
            if (i >= j) break;
        }
        if (cf.glen[ci] < 0) break;
        g += cf.freq[ci] * df.freq[di];
    }
    return g;
}
// newgap(cf, dlc, j), gfreq.cc:523-532
__device__ double newgap_cj(const SList cf, const DList dlc, int j)
{
    for (int ci = 0; cf.glen[ci] >= 0; ++ci) {
        int i = gaplen_sd(cf.glen[ci], dlc);
        if (i >= j) return cf.freq[ci];
    }
    return 0;
}
// newgap(df, i, dld), gfreq.cc:534-545
__device__ double newgap_di(const SList df, int i, const DList dld)
{
    double g = 0;
    int k = 0;
    for (int di = 0; df.glen[di] >= 0; ++di) {
        while (df.glen[di] >= dld.glen(k + 1)) ++k;
        if (i < df.glen[di] + dld.nins(k)) break;
        g += df.freq[di];
    }
    return g;
}
// cleardelta / copydelta, gfreq.cc:551-563
__device__ __forceinline__ void cleardelta(const DList d) { d.set(0, 0, 0); d.set(1, INT_MAX, 0); }
__device__ void copydelta(const DList dst, const DList src)
{
    int k = 0;
    do { dst.set(k, src.glen(k), src.nins(k)); ++k; } while (src.glen(k) < INT_MAX);
    dst.set(k, src.glen(k), src.nins(k));
}
// newdelta(dlt, df, dln, 1), gfreq.cc:570-587; dst may be the same list as src (in-place, as the
// reference does for update(h, h, ...)): every store trails the loads it could affect
__device__ void newdelta(const DList dst, const SList df, const DList src)
{
    int kd = 0, ks = 0;
    int tg = 0, tn = 0;                                   // tmp = ZeroDelta
    for (int di = 0; df.glen[di] >= 0; ++di) {
        int g = df.glen[di];
        if (g >= src.glen(ks)) {
            while (g >= src.glen(ks + 1)) ++ks;
            int sn = src.nins(ks);
            if (sn > tn) {
                dst.set(kd++, tg, tn);
                tn = sn;
                tg = g + 1;
            }
        }
    }
    dst.set(kd++, tg, tn);
    dst.set(kd, INT_MAX, 0);
}
// incdelta(dlt, dln, 1), gfreq.cc:598-605
__device__ void incdelta2(const DList dst, const DList src)
{
    int k = 0;
    do { dst.set(k, src.glen(k), src.nins(k) + 1); ++k; } while (src.glen(k) < INT_MAX);
    dst.set(k, src.glen(k), src.nins(k));
}

// ---- column scorers: PwdM::sim?? (src/maln.h:160-172, src/maln2.cc:534-623,1230-1296) -----------
__device__ double sim2(const DevProb &P, int m, int n)
{
    const DevSide &a = P.a, &b = P.b;
    const uint8_t *ar = res_at(a, m), *br = res_at(b, n);
    const double *mtx = P.simmtx;
    const int dim = P.simdim;
    double s = 0;
    switch (P.sim2_kind) {
    case 0: return 0;
    case 11: return mtx[(size_t) ar[0] * dim + br[0]];
    case 120: for (int j = 0; j < b.many; ++j) s += mtx[(size_t) ar[0] * dim + br[j]]; return s;
    case 121: for (int j = 0; j < b.many; ++j) s += mtx[(size_t) ar[0] * dim + br[j]] * b.weight[j]; return s;
    case 13: return vss_at(b, n)[b.felm + ar[0]];
    case 210: for (int i = 0; i < a.many; ++i) s += mtx[(size_t) br[0] * dim + ar[i]]; return s;
    case 211: for (int i = 0; i < a.many; ++i) s += mtx[(size_t) br[0] * dim + ar[i]] * a.weight[i]; return s;
    case 220:
        for (int i = 0; i < a.many; ++i)
            for (int j = 0; j < b.many; ++j) s += mtx[(size_t) ar[i] * dim + br[j]];
        return s;
    case 221:
        for (int i = 0; i < a.many; ++i) {
            double st = 0;
            for (int j = 0; j < b.many; ++j) st += mtx[(size_t) ar[i] * dim + br[j]] * b.weight[j];
            s += st * a.weight[i];
        }
        return s;
    case 230: { const double *vb = vss_at(b, n) + b.felm;
        for (int i = 0; i < a.many; ++i) s += vb[ar[i]]; return s; }
    case 231: { const double *vb = vss_at(b, n) + b.felm;
        for (int i = 0; i < a.many; ++i) s += vb[ar[i]] * a.weight[i]; return s; }
    case 31: return vss_at(a, m)[a.felm + br[0]];
    case 320: { const double *va = vss_at(a, m) + a.felm;
        for (int j = 0; j < b.many; ++j) s += va[br[j]]; return s; }
    case 321: { const double *va = vss_at(a, m) + a.felm;
        for (int j = 0; j < b.many; ++j) s += va[br[j]] * b.weight[j]; return s; }
    case 33: { const double *va = vss_at(a, m) + a.felm, *vb = vss_at(b, n);
        for (int j = 0; j < b.felm; ++j) s += va[j] * vb[j]; return s; }
    case 330: { const double *va = vss_at(a, m) + a.felm, *vb = vss_at(b, n);
        const int dc[6] = {0, 1, 2, 3, 5, 9};                        // decompact, mseq.h:41
        for (int j = 0; j < b.felm; ++j) s += va[dc[j]] * vb[j]; return s; }
    }
    return 0;
}

// unp1, maln.h:185-187
__device__ __forceinline__ double unpa(const DevProb &P, int m, int n) { return thk_at(P.a, m)[0] * thk_at(P.b, n)[2] * -P.u; }
__device__ __forceinline__ double unpb(const DevProb &P, int n, int m) { return thk_at(P.b, n)[0] * thk_at(P.a, m)[2] * -P.u; }

// ---- naive engine gap-open cost: PwdM::crg?? (maln2.cc:881-1024, 1454-1614) ---------------------
__device__ double crg2(const DevProb &P, const int *gla, const int *glb, int st, int m, int n, int d3)
{   // gla[i*st], glb[j*st]: member running gap lengths of the record
    const DevSide &a = P.a, &b = P.b;
    const uint8_t *as = res_at(a, m), *bs = res_at(b, n);
    const double *agd = a.gapdens + (size_t)(m + 1) * a.many, *apg = a.postgapdens + (size_t)(m + 1) * a.many;
    const double *bgd = b.gapdens + (size_t)(n + 1) * b.many, *bpg = b.postgapdens + (size_t)(n + 1) * b.many;
    const double *wa = a.weight, *wb = b.weight;
    const int an = a.many, bn = b.many;
#define GLA(i) gla[(size_t)(i) * st]
#define GLB(j) glb[(size_t)(j) * st]
#define NG(c) ((c) > 1)
    double g = 0;
    switch (P.crg2_kind) {
    case 11:
        if (d3 == 0) {
            if (NG(as[0]) && bgd[0] > 0 && GLA(0) >= GLB(0)) return bgd[0] * P.basic_gop;
            if (NG(bs[0]) && agd[0] > 0 && GLB(0) >= GLA(0)) return agd[0] * P.basic_gop;
        } else if (d3 > 0) {
            if (bpg[0] > 0 && GLA(0) >= GLB(0)) return bpg[0] * P.basic_gop;
        } else {
            if (apg[0] > 0 && GLB(0) >= GLA(0)) return apg[0] * P.basic_gop;
        }
        return 0;
    case 120: case 121: {
        const bool w = P.crg2_kind & 1;
        if (d3 == 0) {
            if (NG(as[0])) {
                for (int j = 0; j < bn; ++j) if (bgd[j] > 0 && GLA(0) >= GLB(j)) g += w ? wb[j] * bgd[j] : bgd[j];
            } else if (agd[0] > 0) {
                for (int j = 0; j < bn; ++j) if (NG(bs[j]) && GLB(j) >= GLA(0)) g += w ? wb[j] * agd[0] : agd[0];
            }
        } else if (d3 > 0) {
            if (NG(as[0]))
                for (int j = 0; j < bn; ++j) if (bpg[j] > 0 && GLA(0) >= GLB(j)) g += w ? wb[j] * bpg[j] : bpg[j];
        } else if (apg[0] > 0) {
            for (int j = 0; j < bn; ++j) if (NG(bs[j]) && GLB(j) >= GLA(0)) g += w ? wb[j] * apg[0] : apg[0];
        }
        return g * P.basic_gop;
    }
    case 210: case 211: {
        const bool w = P.crg2_kind & 1;
        if (d3 == 0) {
            if (NG(bs[0])) {
                for (int i = 0; i < an; ++i) if (agd[i] > 0 && GLB(0) >= GLA(i)) g += w ? wa[i] * agd[i] : agd[i];
            } else if (bgd[0] > 0) {
                for (int i = 0; i < an; ++i) if (NG(as[i]) && GLA(i) >= GLB(0)) g += w ? wa[i] * bgd[0] : bgd[0];
            }
        } else if (d3 < 0) {
            if (NG(bs[0]))
                for (int i = 0; i < an; ++i) if (apg[i] > 0 && GLB(0) >= GLA(i)) g += w ? wa[i] * apg[i] : apg[i];
        } else if (!w || (bs[0] && bs[1])) {                 // crg21w :1518 tests `*bs && bs[1]`
            if (bpg[0] > 0)
                for (int i = 0; i < an; ++i) if (NG(as[i]) && GLA(i) >= GLB(0)) g += w ? wa[i] * bpg[0] : bpg[0];
        }
        return g * P.basic_gop;
    }
    case 220:
        if (d3 == 0) {
            for (int i = 0; i < an; ++i) {
                if (NG(as[i])) {
                    for (int j = 0; j < bn; ++j) if (bgd[j] > 0 && GLA(i) >= GLB(j)) g += bgd[j];
                } else if (agd[i] > 0) {
                    for (int j = 0; j < bn; ++j) if (NG(bs[j]) && GLB(j) >= GLA(i)) g += agd[i];
                }
            }
        } else if (d3 > 0) {
            for (int i = 0; i < an; ++i)
                if (NG(as[i]))
                    for (int j = 0; j < bn; ++j) if (bpg[j] > 0 && GLA(i) >= GLB(j)) g += bpg[j];
        } else {
            for (int j = 0; j < bn; ++j)
                if (NG(bs[j]))
                    for (int i = 0; i < an; ++i) if (apg[i] > 0 && GLB(j) >= GLA(i)) g += apg[i];
        }
        return g * P.basic_gop;
    case 221:
        if (d3 == 0) {
            for (int i = 0; i < an; ++i) {
                double s = 0;
                if (NG(as[i])) {
                    for (int j = 0; j < bn; ++j) if (bgd[j] > 0 && GLA(i) >= GLB(j)) s += wb[j] * bgd[j];
                } else if (agd[i] > 0) {
                    for (int j = 0; j < bn; ++j) if (NG(bs[j]) && GLB(j) >= GLA(i)) s += wb[j] * agd[i];
                }
                g += s * wa[i];
            }
        } else if (d3 > 0) {
            for (int i = 0; i < an; ++i)
                if (NG(as[i])) {
                    double s = 0;
                    for (int j = 0; j < bn; ++j) if (bpg[j] > 0 && GLA(i) >= GLB(j)) s += wb[j] * bpg[j];
                    g += s * wa[i];
                }
        } else {
            for (int j = 0; j < bn; ++j)
                if (NG(bs[j])) {
                    double s = 0;
                    for (int i = 0; i < an; ++i) if (apg[i] > 0 && GLB(j) >= GLA(i)) s += wa[i] * apg[i];
                    g += s * wb[j];
                }
        }
        return g * P.basic_gop;
    }
#undef GLA
#undef GLB
#undef NG
    return 0;
}

// ---- Fwd2c<recd_t>::gapopen (src/fwd2c.cc:52-91, :107-111, :152-160, :203-212) ------------------
template <int KIND>
__device__ double gapopen(const DevProb &P, Rec rc, int m, int n, int d3)
{
    if (KIND == 0) {                                    // DPunit, no DiThk (quick mode is off-path)
        double axb = 0;
        const int dr = dir_of(P, rc);
        if (d3 > 0) { if (!isvert(dr)) axb = thk_at(P.a, m)[0] * thk_at(P.b, n)[2]; }
        else if (d3 < 0) { if (!ishori(dr)) axb = thk_at(P.b, n)[0] * thk_at(P.a, m)[2]; }
        else return 0;
        return P.basic_gop * axb;                       // vgop, maln.h:320
    } else if (KIND == 1) {                             // _hf: newgap1 / newgap2, maln.h:296-308
        const DList dla = dla_of(P, rc);
        const int glb = P.glb[rc.x][rc.i];
        if (d3 > 0) {
            const SList acf = gfq_at(P.a, 0, m);
            if (acf.glen[0] < 0) return 0;
            if (acf.glen[1] >= 0) return P.weighted_gop * newgap_cj(acf, dla, glb);
            return (dla.nins(0) + acf.glen[0] >= glb) ? (P.weighted_gop * acf.freq[0]) : 0;
        } else {
            const SList adf = gfq_at(P.a, d3 == 0 ? 1 : 2, m);
            if (adf.glen[0] < 0) return 0;
            if (adf.glen[1] >= 0) return P.weighted_gop * newgap_di(adf, glb, dla);
            return (glb >= dla.nins(0) + adf.glen[0]) ? (P.weighted_gop * adf.freq[0]) : 0;
        }
    } else if (KIND == 2) {                             // _pf: newgap3, maln.h:316-319
        const DList dla = dla_of(P, rc), dlb = dlb_of(P, rc);
        if (d3 == 0)
            return newgap4(gfq_at(P.a, 0, m), dla, gfq_at(P.b, 1, n), dlb) * P.basic_gop
                 + newgap4(gfq_at(P.b, 0, n), dlb, gfq_at(P.a, 1, m), dla) * P.basic_gop;
        else if (d3 > 0)
            return newgap4(gfq_at(P.a, 0, m), dla, gfq_at(P.b, 2, n), dlb) * P.basic_gop;
        else
            return newgap4(gfq_at(P.b, 0, n), dlb, gfq_at(P.a, 2, m), dla) * P.basic_gop;
    } else {                                            // _nv
        const int *gl = P.glb[rc.x] + rc.i;
        return crg2(P, gl, gl + (size_t) P.a.many * P.width, P.width, m, n, d3);
    }
}

// ---- Fwd2c<recd_t>::update (src/fwd2c.cc:94-102, :114-128, :163-181, :215-233) ------------------
template <int KIND>
__device__ void update(const DevProb &P, Rec dst, Rec src, int m, int n, double gpn, int d3)
{
    const int sd = dir_of(P, src);
    int dir;
    if (d3 > 0) dir = ishori(sd) ? D_NEWV : D_VERT;
    else if (d3 < 0) dir = isvert(sd) ? D_NEWH : D_HORI;
    else dir = isdiag(sd) ? D_DIAG : D_NEWD;
    if (KIND == 1) {
        if (d3 == 0) { newdelta(dla_of(P, dst), gfq_at(P.a, 1, m), dla_of(P, src)); P.glb[dst.x][dst.i] = 0; }
        else if (d3 > 0) {
            int g = P.glb[src.x][src.i] + 1;
            newdelta(dla_of(P, dst), gfq_at(P.a, 1, m), dla_of(P, src));
            P.glb[dst.x][dst.i] = g;
        } else { incdelta2(dla_of(P, dst), dla_of(P, src)); P.glb[dst.x][dst.i] = 0; }
    } else if (KIND == 2) {
        if (d3 == 0) {
            newdelta(dla_of(P, dst), gfq_at(P.a, 1, m), dla_of(P, src));
            newdelta(dlb_of(P, dst), gfq_at(P.b, 1, n), dlb_of(P, src));
        } else if (d3 > 0) {
            newdelta(dla_of(P, dst), gfq_at(P.a, 1, m), dla_of(P, src));
            incdelta2(dlb_of(P, dst), dlb_of(P, src));
        } else {
            newdelta(dlb_of(P, dst), gfq_at(P.b, 1, n), dlb_of(P, src));
            incdelta2(dla_of(P, dst), dla_of(P, src));
        }
    } else if (KIND == 3) {                             // elongap, mgaps.cc:442-451
        const int st = P.width, an = P.a.many, bn = P.b.many;
        int *dg = P.glb[dst.x] + dst.i;
        const int *sg = P.glb[src.x] + src.i;
        const uint8_t *as = res_at(P.a, m), *bs = res_at(P.b, n);
        for (int i = 0; i < an; ++i) {
            int pr = sg[(size_t) i * st];
            dg[(size_t) i * st] = (d3 >= 0) ? ((as[i] <= 1) ? pr + 1 : 0) : pr + 1;
        }
        for (int j = 0; j < bn; ++j) {
            int pr = sg[(size_t)(an + j) * st];
            dg[(size_t)(an + j) * st] = (d3 <= 0) ? ((bs[j] <= 1) ? pr + 1 : 0) : pr + 1;
        }
    }
    const double v = val_of(P, src) + gpn;
    P.dir[dst.x][dst.i] = (uint8_t) dir;
    val_of(P, dst) = v;
}

// dpunit.cc: reset() -> black record ; copy()
template <int KIND>
__device__ void rec_reset(const DevProb &P, Rec r)
{
    val_of(P, r) = NEVSEL;
    P.dir[r.x][r.i] = 0;
    if (KIND == 1) { P.glb[r.x][r.i] = 0; cleardelta(dla_of(P, r)); }
    else if (KIND == 2) { cleardelta(dla_of(P, r)); cleardelta(dlb_of(P, r)); }
    else if (KIND == 3) {
        const int tot = P.a.many + P.b.many;
        for (int k = 0; k < tot; ++k) P.glb[r.x][r.i + (size_t) k * P.width] = 0;
    }
}
template <int KIND>
__device__ void rec_copy(const DevProb &P, Rec dst, Rec src)
{
    val_of(P, dst) = val_of(P, src);
    P.dir[dst.x][dst.i] = P.dir[src.x][src.i];
    if (KIND == 1) { copydelta(dla_of(P, dst), dla_of(P, src)); P.glb[dst.x][dst.i] = P.glb[src.x][src.i]; }
    else if (KIND == 2) { copydelta(dla_of(P, dst), dla_of(P, src)); copydelta(dlb_of(P, dst), dlb_of(P, src)); }
    else if (KIND == 3) {
        const int tot = P.a.many + P.b.many;
        for (int k = 0; k < tot; ++k) P.glb[dst.x][dst.i + (size_t) k * P.width] = P.glb[src.x][src.i + (size_t) k * P.width];
    }
}

// ---- Fwd2c::initB, src/fwd2c.h:138-176 : the two boundary chains -----------------------------------
// The chains are sequential by nature (corner k needs corner k-1) but each corner is only needed one
// anti-diagonal before its first reader, so they are folded into the sweep: during step d one thread
// advances the top chain to corner (a.left, d+1-a.left), another the left chain to corner
// (d+1-b.left, b.left).  The running chain records live in their own slots (XBT / XBL, ping-pong),
// because the H slot a corner is copied to is consumed in place by the diagonal update of its reader.
template <int KIND>
__device__ void top_step(const DevProb &P, int n)           // corner (a.left, n), b.left < n <= a.left + rr
{
    const DevSide &a = P.a, &b = P.b;
    const int ai = a.left - 1, bi = n - 1;                   // bsi sits on column n-1 (fwd2c.h:151-159)
    const int k = n - b.left;
    const Rec cur = {XBT, k & 1}, prv = {XBT, (k - 1) & 1}, h = {XH, (n - a.left) - (P.lw - 1)};
    const double pub = unpb(P, bi, ai);
    double gnp = gapopen<KIND>(P, prv, ai, bi, -1);
    // (initA's loop variable is the COLUMN the step consumes, one less than the corner it produces: its long-gap switch comes one
    //  corner later than initB's, fwd2c.h:128-133 against :151-158)
    gnp = (n - P.rect - b.left < P.codonk1) ? gnp + pub : (P.v2divv1 * gnp + P.u2divu1 * pub);
    update<KIND>(P, cur, prv, ai, bi, gnp, -1);
    rec_copy<KIND>(P, h, cur);
}
template <int KIND>
__device__ void left_step(const DevProb &P, int m)          // corner (m, b.left), a.left < m <= b.left - rr
{
    const DevSide &a = P.a, &b = P.b;
    // forwardA (rect) advances its left boundary at the start of every row with b's iterator wherever the previous row left it
    // -- position 0 before the first row, b.right afterwards (the reference never resets it, fwd2c.h:244-252) -- and without the
    // long-gap switch
    const int ai = m - 1, bi = P.rect ? (ai == a.left ? 0 : b.right) : b.left - 1;
    const int k = m - a.left;
    const Rec cur = {XBL, k & 1}, prv = {XBL, (k - 1) & 1}, h = {XH, (b.left - m) - (P.lw - 1)};
    const double pua = unpa(P, ai, bi);
    double gnp = gapopen<KIND>(P, prv, ai, bi, 1);
    gnp = (P.rect || m - a.left < P.codonk1) ? gnp + pua : (P.v2divv1 * gnp + P.u2divu1 * pua);
    update<KIND>(P, cur, prv, ai, bi, gnp, 1);
    rec_copy<KIND>(P, h, cur);
}

// ---- one DP cell: the body of the n-loop of forwardB, src/fwd2c.h:393-471 ------------------------
template <int KIND, bool NOLL3>
__device__ void cell(const DevProb &P, int m, int n, uint8_t *tr)
{
    const DevSide &a = P.a, &b = P.b;
    const int i = (n - m) - (P.lw - 1);
    const Rec h = {XH, i}, hu = {XH, i + 1}, hl = {XH, i - 1};
    const Rec g = {XG, i}, gu = {XG, i + 1}, g2 = {XG2, i}, g2u = {XG2, i + 1};
    const Rec f = {XF, i}, fl = {XF, i - 1}, f2 = {XF2, i}, f2l = {XF2, i - 1};
    int bits = 0;
    // diagonal
    const double dab = sim2(P, m, n);
    double gop = gapopen<KIND>(P, h, m, n, 0);
    update<KIND>(P, h, h, m, n, dab + gop, 0);
    Rec mx = g;
    int sel2 = 0;
    // forwardA (P.rect): the first row may open a vertical gap from the top boundary and the first column a horizontal one from the
    // left boundary (forwardB skips both), unpa is evaluated per cell, and Vertical2 opens at v2divv1 + gop (fwd2c.h:276)
    if (m > a.left || P.rect) {
        // vertical; pua is evaluated once per row at the row's first column unless a.inex.nils
        // (fwd2c.h:380,402)
        int nf = m + P.lw; if (nf < b.left) nf = b.left;
        const double pua = unpa(P, m, (a.nils || P.rect) ? n : nf);
        double gnp = gapopen<KIND>(P, gu, m, n, 1);
        gop = gapopen<KIND>(P, hu, m, n, 1);
        const bool hu_nv = !isvert(dir_of(P, hu));
        if (hu_nv && (val_of(P, hu) + gop > val_of(P, gu) + gnp)) update<KIND>(P, g, hu, m, n, gop, 1);
        else { update<KIND>(P, g, gu, m, n, gnp, 1); bits |= T_GEXT; }
        val_of(P, g) += pua;
        if (NOLL3) {
            gnp = P.v2divv1 * gapopen<KIND>(P, g2u, m, n, 1);
            gop = P.rect ? P.v2divv1 + gop : P.v2divv1 * gop;
            if (hu_nv && (val_of(P, hu) + gop > val_of(P, g2u) + gnp)) update<KIND>(P, g2, hu, m, n, gop, 1);
            else { update<KIND>(P, g2, g2u, m, n, gnp, 1); bits |= T_G2EXT; }
            val_of(P, g2) += P.u2divu1 * pua;
            if (val_of(P, g2) > val_of(P, mx)) { mx = g2; sel2 = 1; }
        }
    }
    if (n > b.left || P.rect) {
        // horizontal: F(m, n-1) lives at diagonal index r-1 (the reference carries it in the scalar f1)
        const double pub = unpb(P, n, m);
        double gnp = gapopen<KIND>(P, fl, m, n, -1);
        gop = gapopen<KIND>(P, hl, m, n, -1);
        const bool hl_nh = !ishori(dir_of(P, hl));
        if (hl_nh && (val_of(P, hl) + gop > val_of(P, fl) + gnp)) update<KIND>(P, f, hl, m, n, gop, -1);
        else { update<KIND>(P, f, fl, m, n, gnp, -1); bits |= T_FEXT; }
        val_of(P, f) += pub;
        if (val_of(P, f) >= val_of(P, mx)) { mx = f; sel2 = 0; }
        if (NOLL3) {
            gnp = P.v2divv1 * gapopen<KIND>(P, f2l, m, n, -1);
            gop = P.v2divv1 * gop;
            if (hl_nh && (val_of(P, hl) + gop > val_of(P, f2l) + gnp)) update<KIND>(P, f2, hl, m, n, gop, -1);
            else { update<KIND>(P, f2, f2l, m, n, gnp, -1); bits |= T_F2EXT; }
            val_of(P, f2) += P.u2divu1 * pub;
            if (val_of(P, f2) >= val_of(P, mx)) { mx = f2; sel2 = 1; }
        }
    }
    if (P.nbonus) {                                      // intron-position bonus, fwd2c.h:446-452 (table: g2g_engine.hip)
        for (int k = 0; k < P.nbonus; ++k)
            if (P.bon_m[k] == m && P.bon_n[k] == n) { val_of(P, h) += P.bon_h[k]; val_of(P, mx) += P.bon_mx[k]; break; }
    }
    // diagonal wins ties (fwd2c.h:453)
    if (val_of(P, mx) > val_of(P, h)) { rec_copy<KIND>(P, h, mx); if (sel2) bits |= T_SEL2; }
    *tr = (uint8_t)(bits | dir2code(dir_of(P, h)));
}

template <int KIND, bool NOLL3>
__device__ void run_forward(const DevProb &P)
{
    const DevSide &a = P.a, &b = P.b;
    const int tid = threadIdx.x, nt = blockDim.x;
    // initializeC: every record black (fwd2c.cc:33-40,131-147,184-198,236-251)
    for (int i = tid; i < P.width; i += nt) {
        for (int x = 0; x < NX; ++x) {
            if (!NOLL3 && (x == XG2 || x == XF2)) continue;
            Rec r = {x, i};
            rec_reset<KIND>(P, r);
        }
    }
    if (tid < 4) {                                      // chain slots
        Rec r = {tid < 2 ? XBT : XBL, tid & 1};
        rec_reset<KIND>(P, r);
    }
    __syncthreads();
    if (tid == 0) {                                     // origin, fwd2c.h:145-149
        Rec h = {XH, (b.left - a.left) - (P.lw - 1)};
        const int odir = P.rect ? 0 : D_DIAG;           // initA clears the origin (direction DEAD, fwd2c.h:116), initB sets DIAG
        val_of(P, h) = 0; P.dir[XH][h.i] = odir;
        val_of(P, Rec{XBT, 0}) = 0; P.dir[XBT][0] = odir;
        val_of(P, Rec{XBL, 0}) = 0; P.dir[XBL][0] = odir;
    }
    __syncthreads();
    int rrt = b.right - a.left; if (P.up < rrt) rrt = P.up;          // last top-chain diagonal
    int rrl = b.left - a.right; if (P.lw > rrl) rrl = P.lw;          // last left-chain diagonal
    const int n_top_last = a.left + rrt, m_left_last = b.left - rrl;
    // forwardA's cells READ the boundary records forwardB's never look at -- the corner above a first-row cell, the corner left of a
    // first-column cell -- so in rect mode the chains run one corner further ahead: the first corner of each before the sweep, then
    // corner k + 2 during step k (the slot it lands in has neither reader nor writer on that anti-diagonal)
    const int ahead = P.rect ? 1 : 0;
    if (ahead) {
        if (tid == 0 && b.left + 1 <= n_top_last) top_step<KIND>(P, b.left + 1);
        if (tid == 1 % nt && a.left + 1 <= m_left_last) left_step<KIND>(P, a.left + 1);
        __syncthreads();
    }
    for (int d = P.d0; d <= P.d1; ++d) {
        int mlo, mhi;
        diag_rows(d, a.left, a.right, b.left, b.right, P.lw, P.up, &mlo, &mhi);
        uint8_t *trow = P.trace + (size_t)(d - P.d0) * P.tstride;
        for (int m = mlo + tid; m <= mhi; m += nt) cell<KIND, NOLL3>(P, m, d - m, trow + (m - mlo));
        // boundary corners first read on anti-diagonal d + 1 (given to the threads with the fewest cells)
        const int cnt = mhi >= mlo ? mhi - mlo + 1 : 0;
        const int nb = d + 1 + ahead - a.left, mb = d + 1 + ahead - b.left;
        if (tid == (cnt + nt - 1) % nt && nb > b.left && nb <= n_top_last) top_step<KIND>(P, nb);
        if (tid == (cnt + nt - 2) % nt && mb > a.left && mb <= m_left_last) left_step<KIND>(P, mb);
        __syncthreads();
    }
    if (tid == 0) {
        Rec e = {XH, (b.right - a.right) - (P.lw - 1)};
        *P.score = val_of(P, e);
    }
}

#ifdef G2G_TU_V1
extern "C" __global__ void __launch_bounds__(G2G_FWD_THREADS)
g2g_forward_kernel(const DevProb *probs, const int *idx)
{
    const DevProb &P = probs[idx[blockIdx.x]];
    if (P.kind < 0) return;                             // rejected by the host-side argument check
    if (P.noll == 3) {
        switch (P.kind) {
        case 0: run_forward<0, true>(P); break;
        case 1: run_forward<1, true>(P); break;
        case 2: run_forward<2, true>(P); break;
        default: run_forward<3, true>(P); break;
        }
    } else {
        switch (P.kind) {
        case 0: run_forward<0, false>(P); break;
        case 1: run_forward<1, false>(P); break;
        case 2: run_forward<2, false>(P); break;
        default: run_forward<3, false>(P); break;
        }
    }
}
#else
extern "C" __global__ void g2g_forward_kernel(const DevProb *probs, const int *idx);
#endif

// ---- backtrack: rebuild what Vmf::traceback(-1) returns (src/vmf.cc:105-120) --------------------
// The reference appends a record {m, n, ptr} whenever the final H of a cell has dir NEWD/NEWV/NEWH
// (fwd2c.h:465-467) and chains records through `ptr`, which update()/copy() propagate from the source
// record.  Walking the trace bytes from the end corner visits exactly the H-level cells of that chain.
__device__ __forceinline__ uint8_t trace_at(const DevProb &P, int m, int n)
{
    int mlo, mhi;
    const int d = m + n;
    diag_rows(d, P.a.left, P.a.right, P.b.left, P.b.right, P.lw, P.up, &mlo, &mhi);
    return P.trace[(size_t)(d - P.d0) * P.tstride + (m - mlo)];
}

// Backtrack: ONE WAVE per DP.  The walk itself is a dependent chain (each trace byte decides which byte is read
// next), so what matters is the latency per step: the wave keeps a window of the trace in LDS -- 64 anti-diagonals
// below the current cell x 64 rows around the expected path -- filled by all 64 lanes at once (lane L copies its
// diagonal's segment), and lane 0 walks inside it until the path leaves the window.  (One lane per DP with 64 DPs
// in a wave took 30 ms per sweep: every step was an HBM/L2 round trip, serialised over divergent lanes.)
#define TB_W 64
#ifdef G2G_TU_V1
extern "C" __global__ void __launch_bounds__(64) g2g_traceback_kernel(const DevProb *probs, int nprob)
{
    __shared__ uint8_t win[TB_W][TB_W + 4];
    __shared__ int wbase[TB_W];
    const int ip = blockIdx.x;
    if (ip >= nprob) return;
    const DevProb &P = probs[ip];
    if (P.kind < 0) return;
    const int lane = threadIdx.x;
    const int al = P.a.left, ar = P.a.right, bl = P.b.left, br = P.b.right, lw = P.lw, up = P.up;
    const int d0 = P.d0, tstride = P.tstride;
    const uint8_t *trace = P.trace;
    int2 *out = P.otrace;
    const int tcap = P.tcap;
    // walker state (meaningful in lane 0, broadcast at every window change)
    int m = ar - 1, n = br - 1, cnt = 0, state = 0, ext = 0, fin = 0, bad = 0;
    int rr0 = bl - al, rr0_set = 0;
    if (lane == 0) out[cnt++] = make_int2(ar, br);       // fwd2c.h:476
    int budget = 4 * (ar - al + br - bl) + 16;
    while (!fin) {
        // ---- window: diagonals dtop, dtop-1, ..., dtop-63; rows around m - L/2 (a diagonal move changes d by 2) ----
        const int dtop = m + n;
        {
            const int d = dtop - lane;
            const int mb = m - (lane + 1) / 2 - TB_W / 2 + 4;
            wbase[lane] = mb;
            if (d >= d0) {
                int mlo, mhi;
                diag_rows(d, al, ar, bl, br, lw, up, &mlo, &mhi);
                const uint8_t *src = trace + (size_t) (d - d0) * tstride;
                for (int j = 0; j < TB_W; ++j) {
                    const int mm = mb + j;
                    win[lane][j] = (mm >= mlo && mm <= mhi) ? src[mm - mlo] : (uint8_t) 0;
                }
            }
        }
        __syncthreads();
        if (lane == 0) {
            for (;;) {
                if (--budget < 0) { fin = 1; bad = 1; break; }                 // longer than any path: corrupt trace
                if (state == 0 && (m < al || n < bl)) { fin = 1; break; }      // reached an initB boundary corner
                const int L = dtop - (m + n);
                if (L >= TB_W) break;
                const int j = m - wbase[L];
                if (j < 0 || j >= TB_W) break;
                const uint8_t t = win[L][j];
                if (state == 0) {
                    const int dc = t & T_DIRMASK;
                    if (m == al && !rr0_set) { rr0 = n - m; rr0_set = 1; }     // `else if (m == a->left) h->ptr = n - m`
                    if ((dc == 2 || dc == 4 || dc == 6) && cnt < tcap - 1) out[cnt++] = make_int2(m, n);
                    if (dc == 1 || dc == 2) { --m; --n; }
                    else if (dc == 3 || dc == 4) { ext = (t & T_SEL2) ? T_G2EXT : T_GEXT; state = 1; }   // H copied G or G2
                    else if (dc == 5 || dc == 6) { ext = (t & T_SEL2) ? T_F2EXT : T_FEXT; state = 2; }
                    else { fin = 1; bad = 1; break; }                          // never written: corrupt
                }
                if (state == 1) {                                              // walk the vertical run
                    --m;
                    if (!(t & ext) || m < al) state = 0;
                } else if (state == 2) {
                    --n;
                    if (!(t & ext) || n < bl) state = 0;
                }
            }
        }
        __syncthreads();
        m = __shfl(m, 0); n = __shfl(n, 0); fin = __shfl(fin, 0);
    }
    if (lane == 0) {
        out[cnt++] = make_int2(al, bl);                   // origin record, fwd2c.h:144
        *P.ntrace = bad ? -1 : cnt;                       // -1: the forward pass left a hole (g2g_batch_fetch -> G2G_ERR_DEVICE)
        P.ntrace[1] = rr0;
    }
}
#else
extern "C" __global__ void g2g_traceback_kernel(const DevProb *probs, int nprob);
#endif

// ---- f1: SpScore<recd_t>::calcSkl (reference src/fspscore.h:202-254; calscr src/fspscore.cc:346-363, 472-541) --------
// The sum-of-pairs score of the alignment a standardised skeleton describes, re-evaluated along the path: column scores
// (sim2), unpaired-column penalties and the gap-open COUNTS (raw newgap) of the gap-profile algebra, then
// PwdM::wgop + PwdM::rescale (maln.h:321-325, maln2.cc:245-252).  A dependent chain per alignment (every column
// updates the dynamic gap lists the next one reads): one wave per alignment, lane 0 walks; alignments run in parallel.
// The lists live in the first slot of the v1 state arrays of the problem (free once the forward sweep is over).
struct SpParamsDev { double vab, basic_gep, diffu, diff_u; int flags, reserved; };      // = g2g_spparams
// Gep1st (reference src/mseq.h:355-373, src/mseq.cc:658-758): per member a ring of the last k1 positions that held a
// residue; counts the "long" part of unpaired runs (`lunp`) when Noll = 3.  Rings live in a zeroed workspace in HBM
// ((many) x (k1 + 1) ints per side).  The whole wave executes the chain in lockstep (every lane replays the scalar part:
// same loads, same values, same stores); the member loops are spread over the lanes (member i on lane i mod 64) and the
// weights of the members that count are added IN MEMBER ORDER by a ballot walk every lane performs identically.
__device__ __forceinline__ void gep_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
struct GepDev { int many, k1; int *q, *qp; const double *w; };
__device__ __forceinline__ int gep_shift1(const GepDev &g, int i, int n)
{   // Queue::shift, clib.h:322-326
    int *slot = g.q + (size_t) i * g.k1 + g.qp[i];
    const int old = *slot;
    *slot = n;
    if (++g.qp[i] == g.k1) g.qp[i] = 0;
    return old;
}
__device__ __forceinline__ int gep_oldest(const GepDev &g, int i) { return g.q[(size_t) i * g.k1 + g.qp[i]]; }
__device__ void gep_shift(const GepDev &g, const uint8_t *res, int n)
{   // mseq.cc:675-679
    for (int i = threadIdx.x; i < g.many; i += 64) if (res[i] > 1) gep_shift1(g, i, n);
    gep_wave_sync();
}
__device__ double gep_longup_res(const GepDev &g, const uint8_t *res, int n, int tgl, bool sft)
{   // mseq.cc:681-696
    double lu = 0;
    for (int base = 0; base < g.many; base += 64) {
        const int i = base + (int) threadIdx.x;
        bool hit = false;
        if (i < g.many && res[i] > 1) {
            const int cp = n - (sft ? gep_shift1(g, i, n) : gep_oldest(g, i));
            hit = tgl > cp;
        }
        unsigned long long m = __ballot(hit);
        while (m) {                                            // members that count, ascending: the reference's order of the sum
            const int j = __ffsll((long long) m) - 1;
            m &= m - 1;
            if (g.w) lu += g.w[base + j]; else lu += 1;
        }
    }
    if (sft) gep_wave_sync();
    return lu;
}
__device__ __forceinline__ int sl_len(const SList df) { int k = 0; while (df.glen[k] >= 0) ++k; return k; }
__device__ double gep_longup_half(const GepDev &g, const SList df, const DList dld, int pos)
{   // mseq.cc:729-742
    double lunp = 0;
    for (int k = sl_len(df) - 1; k > 0; --k) {
        const int gi = gaplen_sd(df.glen[k], dld);
        if (gi > pos - gep_oldest(g, 0)) lunp += df.freq[k];
        else break;
    }
    gep_wave_sync();                                           // (every lane has read ring 0 before lane 0 moves it)
    if (threadIdx.x == 0) gep_shift1(g, 0, pos);
    gep_wave_sync();
    return lunp;
}
__device__ double gep_longup_both(const GepDev &g, const SList df, const DList dld, const uint8_t *res, int pos)
{   // mseq.cc:744-758
    double lunp = 0;
    for (int k = sl_len(df) - 1; k > 0; --k) {
        const int gi = gaplen_sd(df.glen[k], dld);
        const double lu = gep_longup_res(g, res, pos, gi, false) * df.freq[k];
        if (lu == 0) break;
        lunp += lu;
    }
    gep_shift(g, res, pos);
    return lunp;
}
// The static lists of the current columns, staged in LDS by the whole wave (one parallel load round) before the merges walk them
// entry by entry: a dependent global load per entry made the chain of a _pf alignment cost ~16 us per column.
#define SP_STAGE 64
struct SpStage { int *gl; double *fq; };                          // 6 slots x SP_STAGE entries in LDS (NULL: read the lists in place)
__device__ __forceinline__ SList sp_staged(const SpStage &G, const DevSide &sd, const int view, const int pos, const int slot)
{
    const int o = sd.off[view][pos + 1], len = sd.off[view][pos + 2] - o;       // entries incl. the terminator
    SList l; l.glen = sd.glen[view] + o; l.freq = sd.freq[view] + o;
    if (!G.gl || len > SP_STAGE) return l;
    int *dg = G.gl + slot * SP_STAGE; double *df = G.fq + slot * SP_STAGE;
    for (int k = threadIdx.x & 63; k < len; k += 64) { dg[k] = l.glen[k]; df[k] = l.freq[k]; }
    l.glen = dg; l.freq = df;
    return l;
}
__device__ __forceinline__ void sp_stage_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// ---- the walk's position-only inputs as a stream ------------------------------------------------------------------------
// What a column of the path needs that does NOT depend on the walk's state -- its score (sim2 / unpa / unpb) and the static
// lists of its positions -- is laid out by a parallel pre-pass (g2g_spprep_kernel: one thread per path column) as one 512-byte
// slot per column, in path order.  The walking wave then reads its inputs through one LDS chunk of SPS_CH slots, fetched with
// full-width loads a chunk ahead, instead of paying several dependent global round trips per column (3.8 us per column before;
// the merges themselves are 0.3 us of instruction issue).  Lists longer than SPS_W entries are marked and read in place.
#define SPS_W 8                       // entries of a packed list, terminator included
#define SPS_CH 16                     // slots per LDS chunk
struct SpSlot { double cs; int len[5]; int pad; int glen[5][SPS_W]; double freq[5][SPS_W]; };   // 512 bytes; len: entries, 0 unused, -1 read in place
struct SpStream { const SpSlot *g; SpSlot *lds; int ncols; int4 r[8]; };
// which list sits where: 0 a.s  1 a.t  2 b.s (_pf) / a.r (_hf)  3 b.t  4 the other side's r (_pf)
__device__ __forceinline__ void sps_fetch(SpStream &S, const int j)
{
    const int4 *src = (const int4 *) (S.g + (size_t) j * SPS_CH);
    const int n16 = min(SPS_CH, S.ncols - j * SPS_CH) * (int) (sizeof(SpSlot) / 16);
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int i = q * 64 + (int) (threadIdx.x & 63); S.r[q] = i < n16 ? src[i] : make_int4(0, 0, 0, 0); }
}
__device__ __forceinline__ const SpSlot *sps_col(SpStream &S, const int c)
{
    if ((c & (SPS_CH - 1)) == 0) {                    // first column of a chunk: every read of the previous chunk is behind us
        const int j = c / SPS_CH;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        int4 *dst = (int4 *) S.lds;
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q * 64 + (int) (threadIdx.x & 63)] = S.r[q];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        if ((j + 1) * SPS_CH < S.ncols) sps_fetch(S, j + 1);
    }
    return S.lds + (c & (SPS_CH - 1));
}
__device__ __forceinline__ SList sps_list(const SpSlot *sl, const int l, const DevSide &sd, const int view, const int pos)
{
    SList r;
    if (sl->len[l] < 0) return gfq_at(sd, view, pos);
    r.glen = sl->glen[l]; r.freq = sl->freq[l];
    return r;
}
// ---- PwdM::stt?? (src/maln2.cc:627-850, 1300-1450): matched / mismatched / unpaired member pairs of a column pair -> FSTAT.
// ha / hb false: the "zero" iterator of a gap segment (res = vss = NULL, thickness {sumwt, 0, sumwt}).  Every lane replays it.
struct SpStat { double mch, mmc, unp; };
__device__ void sp_stt2(const DevProb &P, const int apos, const int bpos, const bool ha, const bool hb, SpStat &S)
{
    const DevSide &a = P.a, &b = P.b;
    const uint8_t *ca = ha ? res_at(a, apos) : (const uint8_t *) 0, *cb = hb ? res_at(b, bpos) : (const uint8_t *) 0;
    const double *va = (ha && a.pseq) ? vss_at(a, apos) : (const double *) 0, *vb = (hb && b.pseq) ? vss_at(b, bpos) : (const double *) 0;
    const double aefq = ha ? thk_at(a, apos)[2] : a.sumwt, befq = hb ? thk_at(b, bpos)[2] : b.sumwt;
    const double van = a.sumwt, vbn = b.sumwt;
    const double *wa = a.weight, *wb = b.weight;
    const int an = a.many, bn = b.many;
    const bool dxd = P.dvsp == 0;
    const uint8_t cmp[17] = {0, 1, 2, 3, 6, 4, 6, 6, 6, 5, 6, 6, 6, 6, 6, 6, 6};       // nccmpctab, seq.cc:32
#define SP_NOTGAP(x) ((x) && (x)[0] > 1)
#define SP_TRUEGAP(x) (!(x) || (x)[0] == 1)
    switch (P.sim2_kind) {
    case G2G_SIM11:
        if (SP_NOTGAP(ca)) {
            if (SP_NOTGAP(cb)) { if (ca[0] == cb[0]) ++S.mch; else ++S.mmc; }
            else if (SP_TRUEGAP(ca)) ++S.unp;
        } else if (SP_NOTGAP(cb) && SP_TRUEGAP(ca)) ++S.unp;
        break;
    case G2G_SIM12I: {
        int sm = 0, g = 0, u = 0;
        if (SP_NOTGAP(ca)) {
            if (cb) {
                for (int j = 0; j < bn; ++j) { if (cb[j] == ca[0]) ++sm; else { if (cb[j] <= 1) ++u; if (cb[j] == 1) ++g; } }
                S.mch += sm; S.mmc += bn - sm - u; S.unp += g;
            } else S.unp += befq;
        } else if (cb && SP_TRUEGAP(ca)) { for (int j = 0; j < bn; ++j) if (cb[j] > 1) ++g; S.unp += g; }
        break; }
    case G2G_SIM12W: {
        double sm = 0, g = 0, u = 0;
        if (SP_NOTGAP(ca)) {
            if (cb) {
                for (int j = 0; j < bn; ++j) { if (cb[j] == ca[0]) sm += wb[j]; else { if (cb[j] <= 1) u += wb[j]; if (cb[j] == 1) g += wb[j]; } }
                S.mch += sm; S.mmc += vbn - sm - u; S.unp += g;
            } else S.unp += befq;
        } else if (cb && SP_TRUEGAP(ca)) { for (int j = 0; j < bn; ++j) if (cb[j] > 1) S.unp += wb[j]; }
        break; }
    case G2G_SIM13:
        if (SP_NOTGAP(ca)) {
            if (vb) {
                int cca = ca[0];
                if (dxd) cca = cmp[cca];
                S.mch += vb[cca]; S.mmc += vbn - vb[1] - vb[cca] - vb[0]; S.unp += vb[1];
            } else S.unp += befq;
        } else if (vb && SP_TRUEGAP(ca)) S.unp += vbn - vb[1] - vb[0];
        break;
    case G2G_SIM21I: {
        int sm = 0, g = 0, u = 0;
        if (SP_NOTGAP(cb)) {
            if (ca) {
                for (int i = 0; i < an; ++i) { if (ca[i] == cb[0]) ++sm; else if (ca[i] <= 1) ++u; if (ca[i] == 1) ++g; }
                S.mch += sm; S.mmc += an - sm - u; S.unp += g;
            } else S.unp += aefq;
        } else if (ca && SP_TRUEGAP(cb)) { for (int i = 0; i < an; ++i) if (ca[i] > 1) ++g; S.unp += g; }
        break; }
    case G2G_SIM21W: {
        double sm = 0, g = 0, u = 0;
        if (SP_NOTGAP(cb)) {
            if (ca) {
                for (int i = 0; i < an; ++i) { if (ca[i] == cb[0]) sm += wa[i]; else if (ca[i] <= 1) u += wa[i]; if (ca[i] == 1) g += wa[i]; }
                S.mch += sm; S.mmc += van - sm - u; S.unp += g;
            } else S.unp += aefq;
        } else if (ca && SP_TRUEGAP(cb)) { for (int i = 0; i < an; ++i) if (ca[i] > 1) S.unp += wa[i]; }
        break; }
    case G2G_SIM22I: {
        int sm = 0, m = 0, g = 0;
        if (ca && cb) {
            for (int j = 0; j < bn; ++j) {
                if (cb[j] > 1) { for (int i = 0; i < an; ++i) { if (ca[i] == cb[j]) ++sm; else if (ca[i] > 1) ++m; if (ca[i] == 1) ++g; } }
                else if (cb[j] == 1) { for (int i = 0; i < an; ++i) if (ca[i] > 1) ++g; }
            }
        } else if (ca) { for (int i = 0; i < an; ++i) if (ca[i] > 1) g += (int) befq; }
        else if (cb) { for (int j = 0; j < bn; ++j) if (cb[j] > 1) g += (int) aefq; }
        S.mch += sm; S.mmc += m; S.unp += g;
        break; }
    case G2G_SIM22W:
        if (ca && cb) {
            for (int j = 0; j < bn; ++j) {
                double sm = 0, g = 0, u = 0;
                if (cb[j] > 1) {
                    for (int i = 0; i < an; ++i) { if (ca[i] == cb[j]) sm += wa[i]; else { if (ca[i] <= 1) u += wa[i]; if (ca[i] == 1) g += wa[i]; } }
                    S.mmc += (van - sm - u) * wb[j]; S.mch += sm * wb[j]; S.unp += g * wb[j];
                } else if (cb[j] == 1) {
                    for (int i = 0; i < an; ++i) if (ca[i] > 1) g += wa[i];
                    S.unp += g * wb[j];
                }
            }
        } else if (ca) { for (int i = 0; i < an; ++i) if (ca[i] > 1) S.unp += befq * wa[i]; }
        else if (cb) { for (int j = 0; j < bn; ++j) if (cb[j] > 1) S.unp += aefq * wb[j]; }
        break;
    case G2G_SIM23I: case G2G_SIM23W: {
        const bool w = P.sim2_kind == G2G_SIM23W;
        if (ca) {
            for (int i = 0; i < an; ++i) {
                int cca = ca[i];
                if (dxd) cca = cmp[cca];
                if (cca > 1) {
                    if (vb) {
                        if (w) { S.mch += vb[cca] * wa[i]; S.mmc += (vbn - vb[1] - vb[cca] - vb[0]) * wa[i]; S.unp += vb[1] * wa[i]; }
                        else { S.mch += vb[cca]; S.mmc += vbn - vb[1] - vb[cca] - vb[0]; S.unp += vb[1]; }
                    } else if ((w ? (int) ca[i] : cca) == 1) S.unp += befq;
                } else if (vb && (w ? (int) ca[i] : cca) == 1) { if (w) S.unp += (vbn - vb[1] - vb[0]) * wa[i]; else S.unp += vbn - vb[1] - vb[0]; }
            }
        } else if (vb) S.unp += aefq * (vbn - vb[1] - vb[0]);
        break; }
    case G2G_SIM31:
        if (SP_NOTGAP(cb)) {
            if (va) {
                int ccb = cb[0];
                if (dxd) ccb = cmp[ccb];
                S.mch += va[ccb]; S.mmc += van - va[1] - va[ccb] - va[0]; S.unp += va[1];
            } else S.unp += aefq;
        } else if (va && SP_TRUEGAP(cb)) S.unp += van - va[1] - va[0];
        break;
    case G2G_SIM32I: case G2G_SIM32W: {
        const bool w = P.sim2_kind == G2G_SIM32W;
        if (cb) {
            for (int j = 0; j < bn; ++j) {
                int ccb = cb[j];
                if (dxd) ccb = cmp[ccb];
                if (ccb > 1) {
                    if (va) {
                        if (w) { S.mch += va[ccb] * wb[j]; S.mmc += (van - va[1] - va[ccb] - va[0]) * wb[j]; S.unp += va[1] * wb[j]; }
                        else { S.mch += va[ccb]; S.mmc += van - va[1] - va[ccb] - va[0]; S.unp += va[1]; }
                    } else if ((w ? (int) cb[j] : ccb) == 1) S.unp += aefq;
                } else if (va && (w ? (int) cb[j] : ccb) == 1) { if (w) S.unp += (van - va[1] - va[0]) * wb[j]; else S.unp += van - va[1] - va[0]; }
            }
        } else if (va) S.unp += befq * (van - va[1] - va[0]);
        break; }
    case G2G_SIM33: case G2G_SIM33N:
        if (va && vb) {
            double sm = 0;
            S.mmc += (van - va[1] - va[0]) * (vbn - vb[1] - vb[0]);
            S.unp += va[1] * (vbn - vb[1] - vb[0]) + (van - va[1] - va[0]) * vb[1];
            const int base = dxd ? 2 : 3, top = a.felm - 2;           // va + base_code .. va + felm - 2 (:834-838, as written)
            for (int k = base; k < top; ++k) sm += va[k] * vb[k];
            S.mch += sm; S.mmc -= sm;
        } else if (va) S.unp += (van - va[1] - va[0]) * befq;
        else if (vb) S.unp += (vbn - vb[1] - vb[0]) * aefq;
        break;
    default: break;
    }
#undef SP_NOTGAP
#undef SP_TRUEGAP
}
template <int KIND, bool STREAM>
__device__ void sp_calscr(const DevProb &P, const SpParamsDev &sp, int mi, int ni, int &apos, int &bpos, int &glb,
                          const DList dla, const DList dlb, double &scr, double &tgap,
                          const bool gep, const GepDev &agep, const GepDev &bgep, double &lunp, SpStat &St, const SpStage &G,
                          SpStream &S, int &c)
{
    const DevSide &a = P.a, &b = P.b;
    // the inputs of the next column: from its slot of the stream (STREAM), or staged from the profiles here
#define SP_SLOT const SpSlot *sl = STREAM ? sps_col(S, c) : (const SpSlot *) 0; ++c
#define SP_LIST(l, side, view, pos, stg) (STREAM ? sps_list(sl, l, side, view, pos) : sp_staged(G, side, view, pos, stg))
#define SP_SYNC if (!STREAM) sp_stage_sync()
    if (KIND == 0) {
        if (mi == ni) {
            while (mi--) { ++apos; ++bpos; scr += sim2(P, apos, bpos); if (!(sp.flags & 1)) sp_stt2(P, apos, bpos, true, true, St); }
        } else if (mi) {
            St.unp += mi;
            const double efq = thk_at(b, bpos)[2];
            tgap += sp.vab * efq;
            double unp = mi * sp.basic_gep;                                       // UnpPenalty, aln.h:275-279
            if (mi > P.codonk1) unp = unp + sp.diffu * (mi - P.codonk1);
            scr += unp * efq;
            apos += mi;
        } else if (ni) {
            St.unp += ni;
            const double efq = thk_at(a, apos)[2];
            tgap += sp.vab * efq;
            double unp = ni * sp.basic_gep;
            if (ni > P.codonk1) unp = unp + sp.diffu * (ni - P.codonk1);
            scr += unp * efq;
            bpos += ni;
        }
    } else if (KIND == 1) {
        if (mi == ni) {
            while (mi--) {
                ++apos; ++bpos;
                SP_SLOT;
                const SList at = SP_LIST(1, a, 1, apos, 1);
                SP_SYNC;
                scr += STREAM ? sl->cs : sim2(P, apos, bpos);
                tgap += newgap_di(at, glb, dla);
                if (!(sp.flags & 1)) sp_stt2(P, apos, bpos, true, true, St);
                newdelta(dla, at, dla);
                if (gep) { lunp += gep_longup_half(bgep, at, dla, bpos); gep_shift(agep, res_at(a, apos), apos); }
                glb = 0;
                SP_SYNC;
            }
        } else if (mi) {
            while (mi--) {
                ++apos;
                SP_SLOT;
                const SList as = SP_LIST(0, a, 0, apos, 0), at = SP_LIST(1, a, 1, apos, 1);
                SP_SYNC;
                scr += STREAM ? sl->cs : unpa(P, apos, bpos);
                tgap += newgap_cj(as, dla, glb);
                if (!(sp.flags & 1)) sp_stt2(P, apos, bpos, true, false, St);
                newdelta(dla, at, dla);
                ++glb;
                if (gep) lunp += gep_longup_res(agep, res_at(a, apos), apos, glb, true);
                SP_SYNC;
            }
        } else if (ni) {
            SList ar; ar.glen = 0; ar.freq = 0;
            if (!STREAM) ar = sp_staged(G, a, 2, apos, 2);      // (apos does not move in this segment)
            SP_SYNC;
            while (ni--) {
                ++bpos;
                SP_SLOT;
                if (STREAM) ar = sps_list(sl, 2, a, 2, apos);              // (every slot of the run carries the list)
                scr += STREAM ? sl->cs : unpb(P, bpos, apos);
                tgap += newgap_di(ar, glb, dla);
                if (!(sp.flags & 1)) sp_stt2(P, apos, bpos, false, true, St);
                incdelta2(dla, dla);
                if (gep) lunp += gep_longup_half(bgep, ar, dla, bpos);
            }
            SP_SYNC;
        }
    } else {
        if (mi == ni) {
            while (mi--) {
                ++apos; ++bpos;
                SP_SLOT;
                const SList as = SP_LIST(0, a, 0, apos, 0), at = SP_LIST(1, a, 1, apos, 1);
                const SList bs = SP_LIST(2, b, 0, bpos, 3), bt = SP_LIST(3, b, 1, bpos, 4);
                SP_SYNC;
                scr += STREAM ? sl->cs : sim2(P, apos, bpos);
                tgap += newgap4(as, dla, bt, dlb)
                      + newgap4(bs, dlb, at, dla);
                if (!(sp.flags & 1)) sp_stt2(P, apos, bpos, true, true, St);
                newdelta(dla, at, dla);
                newdelta(dlb, bt, dlb);
                if (gep) {
                    lunp += gep_longup_both(agep, bt, dlb, res_at(a, apos), apos);
                    lunp += gep_longup_both(bgep, at, dla, res_at(b, bpos), bpos);
                }
                SP_SYNC;
            }
        } else if (mi) {
            SList br; br.glen = 0; br.freq = 0;
            if (!STREAM) br = sp_staged(G, b, 2, bpos, 5);      // (bpos does not move in this segment)
            while (mi--) {
                ++apos;
                SP_SLOT;
                const SList as = SP_LIST(0, a, 0, apos, 0), at = SP_LIST(1, a, 1, apos, 1);
                if (STREAM) br = sps_list(sl, 4, b, 2, bpos);
                SP_SYNC;
                scr += STREAM ? sl->cs : unpa(P, apos, bpos);
                tgap += newgap4(as, dla, br, dlb);
                if (!(sp.flags & 1)) sp_stt2(P, apos, bpos, true, false, St);
                newdelta(dla, at, dla);
                incdelta2(dlb, dlb);
                if (gep) lunp += gep_longup_both(agep, br, dlb, res_at(a, apos), apos);
                SP_SYNC;
            }
        } else if (ni) {
            SList ar; ar.glen = 0; ar.freq = 0;
            if (!STREAM) ar = sp_staged(G, a, 2, apos, 2);
            while (ni--) {
                ++bpos;
                SP_SLOT;
                const SList bs = SP_LIST(2, b, 0, bpos, 3), bt = SP_LIST(3, b, 1, bpos, 4);
                if (STREAM) ar = sps_list(sl, 4, a, 2, apos);
                SP_SYNC;
                scr += STREAM ? sl->cs : unpb(P, bpos, apos);
                tgap += newgap4(bs, dlb, ar, dla);
                if (!(sp.flags & 1)) sp_stt2(P, apos, bpos, false, true, St);
                newdelta(dlb, bt, dlb);
                incdelta2(dla, dla);
                if (gep) lunp += gep_longup_both(bgep, ar, dla, res_at(b, bpos), bpos);
                SP_SYNC;
            }
        }
    }
#undef SP_SLOT
#undef SP_LIST
#undef SP_SYNC
}
// ---- naive units of calcSpScore (NTV modes): SPunit_nv / _w11 / _w22, reference src/fspscore.h:34-78, calcstat
// src/fspscore.cc:60-340, calscr :365-470.  Groups are tiny (2 nj + ni < 8): every lane replays everything.
struct NtvState { int unit; int *gla, *glb; };            // unit: 0 nv, 1 w11, 2 w22, 3 w21; per member running gap lengths
__device__ __forceinline__ bool gep_long1(const GepDev &g, int i, int n, int tgl) { return tgl > n - gep_oldest(g, i); }
__device__ void sp_incrgap(int *gg, const uint8_t *ss, int n)
{   // incrgap, mgaps.cc:431-440
    for (int i = 0; i < n; ++i) { if (!ss || ss[i] <= 1) ++gg[i]; else gg[i] = 0; }
}
__device__ void sp_pregap(const DevSide &sd, int *gl)
{   // Seq::pregap, seq.cc:1870-1883
    for (int i = 0; i < sd.many; ++i) {
        int n = sd.left;
        for ( ; n > 0; --n) if (res_at(sd, n - 1)[i] > 1) break;
        gl[i] = sd.left - n;
    }
}
__device__ void sp_calcstat_ntv(const DevProb &P, const NtvState &N, int d3, int apos, int bpos, double &scr, double &tgap,
                                const bool gep, const GepDev &agep, const GepDev &bgep, double &lunp, SpStat &St)
{
    const DevSide &a = P.a, &b = P.b;
    const int an = a.many, bn = b.many;
    const uint8_t *as = res_at(a, apos), *bs = res_at(b, bpos);
    const double *agd = a.gapdens + (size_t) (apos + 1) * an, *bgd = b.gapdens + (size_t) (bpos + 1) * bn;
    const double *apg = a.postgapdens + (size_t) (apos + 1) * an, *bpg = b.postgapdens + (size_t) (bpos + 1) * bn;
    scr += d3 == 0 ? sim2(P, apos, bpos) : d3 > 0 ? unpa(P, apos, bpos) : unpb(P, bpos, apos);
    if (N.unit == 1) {                                         // SPunit_w11: scalars, member 0 of either side
        const int gla = N.gla[0], glb = N.glb[0];
        if (d3 == 0) {
            const bool ar = as[0] > 1, br = bs[0] > 1;
            const double au = agd[0], bu = bgd[0];
            if (ar && br) { if (as[0] == bs[0]) ++St.mch; else ++St.mmc; }
            else if (ar && bu > 0) {
                St.unp += bu;
                if (glb <= gla) tgap += bu;
                else if (gep) lunp += gep_longup_res(agep, as, apos, glb + 1, true) * bu;
            } else if (au > 0 && br) {
                St.unp += au;
                if (gla <= glb) tgap += au;
                else if (gep) lunp += gep_longup_res(bgep, bs, bpos, gla + 1, true) * au;
            }
        } else if (d3 > 0) {
            if (as[0] > 1) {
                const double bu = thk_at(b, bpos)[2];
                if (bu > 0) {
                    St.unp += bu;
                    if (glb <= gla) tgap += bu;
                    else if (gep) lunp += gep_longup_res(agep, as, apos, glb + 1, true) * bu;
                }
            }
        } else {
            if (bs[0] > 1) {
                const double au = thk_at(a, apos)[2];
                if (au > 0) {
                    St.unp += au;
                    if (gla <= glb) tgap += au;
                    else if (gep) lunp += gep_longup_res(bgep, bs, bpos, gla + 1, true) * au;
                }
            }
        }
        return;
    }
    if (N.unit == 3) {                                         // SPunit_w21 (fspscore.cc:192-250): a several with weights, b a single
        const double *wa = a.weight;
        const bool br = bs[0] > 1;
        const int glb = N.glb[0];
        const double bu = thk_at(b, bpos)[2];
        if (d3 == 0) {
            for (int i = 0; i < an; ++i) {
                const bool ar = as[i] > 1;
                const double au = agd[i];
                if (ar && br) { if (as[i] == bs[0]) St.mch += wa[i]; else St.mmc += wa[i]; }
                else if (ar && bu > 0) {
                    St.unp += wa[i] * bu;
                    if (glb <= N.gla[i]) tgap += wa[i] * bu;
                    else if (gep && gep_long1(agep, i, apos, glb + 1)) lunp += wa[i] * bu;
                } else if (au > 0 && br) {
                    St.unp += wa[i] * au;
                    if (N.gla[i] <= glb) tgap += wa[i] * au;
                    else if (gep && gep_long1(bgep, 0, bpos, N.gla[i] + 1)) lunp += wa[i] * au;
                }
                if (ar && gep) gep_shift1(agep, i, apos);
            }
            if (br && gep) gep_shift1(bgep, 0, bpos);
        } else if (d3 > 0) {
            for (int i = 0; i < an; ++i)
                if (as[i] > 1) {
                    if (bu > 0) {
                        St.unp += wa[i] * bu;
                        if (glb <= N.gla[i]) tgap += wa[i] * bu;
                        else if (gep && gep_long1(agep, i, apos, glb + 1)) lunp += wa[i] * bu;
                    }
                    if (gep) gep_shift1(agep, i, apos);
                }
        } else if (br) {
            for (int i = 0; i < an; ++i) {
                const double au = apg[i];
                if (au > 0) {
                    St.unp += wa[i] * au;
                    if (N.gla[i] <= glb) tgap += wa[i] * au;
                    else if (gep && gep_long1(bgep, 0, bpos, N.gla[i] + 1)) lunp += wa[i] * au;
                }
            }
            if (gep) gep_shift1(bgep, 0, bpos);
        }
        if (gep) gep_wave_sync();
        return;
    }
    const bool w = N.unit == 2;                                // SPunit_w22 weighs the inner sums, SPunit_nv adds them raw
    const double *wta = a.weight, *wtb = b.weight;
    int nmch = 0, nmmc = 0;                                    // SPunit_nv counts locally and adds once (fspscore.cc:63-65,134-138)
    double nunp = 0;
    if (d3 == 0) {
        for (int i = 0; i < an; ++i) {
            const bool ar = as[i] > 1;
            const double au = agd[i];
            double g = 0, l = 0, sm = 0, u = 0;
            for (int j = 0; j < bn; ++j) {
                const bool br = bs[j] > 1;
                const double bu = bgd[j];
                if (ar && br) { if (w) { if (as[i] == bs[j]) sm += wtb[j]; } else { if (as[i] == bs[j]) ++nmch; else ++nmmc; } }
                else if (ar && bu > 0) {
                    if (w) u += wtb[j] * bu; else nunp += bu;
                    if (N.glb[j] <= N.gla[i]) { if (w) g += wtb[j] * bu; else tgap += bu; }
                    else if (gep && gep_long1(agep, i, apos, N.glb[j] + 1)) { if (w) l += wtb[j] * bu; else lunp += bu; }
                } else if (au > 0 && br) {
                    if (w) u += wtb[j] * au; else nunp += au;
                    if (N.gla[i] <= N.glb[j]) { if (w) g += wtb[j] * au; else tgap += au; }
                    else if (gep && gep_long1(bgep, j, bpos, N.gla[i] + 1)) { if (w) l += wtb[j] * au; else lunp += au; }
                }
            }
            if (w) { tgap += g * wta[i]; lunp += l * wta[i]; St.mch += sm * wta[i]; St.mmc += (b.sumwt - sm - u) * wta[i]; St.unp += u * wta[i]; }
            if (ar && gep) gep_shift1(agep, i, apos);
        }
        if (gep) { gep_wave_sync(); gep_shift(bgep, bs, bpos); }
    } else if (d3 > 0) {
        if (w) St.unp += thk_at(b, bpos)[2] * thk_at(a, apos)[0]; else nunp += thk_at(a, apos)[0] * thk_at(b, bpos)[2];
        for (int i = 0; i < an; ++i)
            if (as[i] > 1) {
                double g = 0, l = 0;
                for (int j = 0; j < bn; ++j) {
                    const double bu = bpg[j];
                    if (bu > 0) {
                        if (N.glb[j] <= N.gla[i]) { if (w) g += wtb[j] * bu; else tgap += bu; }
                        else if (gep && gep_long1(agep, i, apos, N.glb[j] + 1)) { if (w) l += wtb[j] * bu; else lunp += bu; }
                    }
                }
                if (w) { tgap += g * wta[i]; lunp += l * wta[i]; }
                if (gep) gep_shift1(agep, i, apos);
            }
    } else {
        if (w) St.unp += thk_at(a, apos)[2] * thk_at(b, bpos)[0]; else nunp += thk_at(a, apos)[2] * thk_at(b, bpos)[0];
        for (int j = 0; j < bn; ++j)
            if (bs[j] > 1) {
                double g = 0, l = 0;
                for (int i = 0; i < an; ++i) {
                    const double au = apg[i];
                    if (au > 0) {
                        if (N.gla[i] <= N.glb[j]) { if (w) g += wta[i] * au; else tgap += au; }
                        else if (gep && gep_long1(bgep, j, bpos, N.gla[i] + 1)) { if (w) l += wta[i] * au; else lunp += au; }
                    }
                }
                if (w) { tgap += g * wtb[j]; lunp += l * wtb[j]; }
                if (gep) gep_shift1(bgep, j, bpos);
            }
    }
    if (!w) { St.mch += nmch; St.mmc += nmmc; St.unp += nunp; }
    if (gep) gep_wave_sync();
}
__device__ void sp_calscr_ntv(const DevProb &P, const NtvState &N, int mi, int ni, int &apos, int &bpos, double &scr, double &tgap,
                              const bool gep, const GepDev &agep, const GepDev &bgep, double &lunp, SpStat &St)
{
    const int an = P.a.many, bn = P.b.many;
    if (mi == ni) {
        while (mi--) {
            ++apos; ++bpos;
            sp_calcstat_ntv(P, N, 0, apos, bpos, scr, tgap, gep, agep, bgep, lunp, St);
            if (N.unit == 1) { N.gla[0] = 0; N.glb[0] = 0; }
            else if (N.unit == 3) { sp_incrgap(N.gla, res_at(P.a, apos), an); N.glb[0] = 0; }
            else { sp_incrgap(N.gla, res_at(P.a, apos), an); sp_incrgap(N.glb, res_at(P.b, bpos), bn); }
        }
    } else if (mi) {
        while (mi--) {
            ++apos;
            sp_calcstat_ntv(P, N, 1, apos, bpos, scr, tgap, gep, agep, bgep, lunp, St);
            if (N.unit == 1) { N.gla[0] = 0; ++N.glb[0]; }
            else if (N.unit == 3) { sp_incrgap(N.gla, res_at(P.a, apos), an); ++N.glb[0]; }
            else { sp_incrgap(N.gla, res_at(P.a, apos), an); sp_incrgap(N.glb, (const uint8_t *) 0, bn); }
        }
    } else if (ni) {
        while (ni--) {
            ++bpos;
            sp_calcstat_ntv(P, N, -1, apos, bpos, scr, tgap, gep, agep, bgep, lunp, St);
            if (N.unit == 1) { ++N.gla[0]; N.glb[0] = 0; }
            else if (N.unit == 3) { sp_incrgap(N.gla, (const uint8_t *) 0, an); N.glb[0] = 0; }
            else { sp_incrgap(N.gla, (const uint8_t *) 0, an); sp_incrgap(N.glb, res_at(P.b, bpos), bn); }
        }
    }
}
// ---- Iiinfo + Iiinfo::StoreIIinfo (src/gsinfo.cc:64-83, 622-684): the intron-position term of SpScore::calcSkl -------------
// Two cursors over the sides' exon-boundary lists; after every diagonal run and every gap of the skeleton the boundaries
// passed so far are compared in the alignment's common coordinate (position + the gaps inserted on that side): each pair
// that coincides adds dns_a x dns_b.  Tiny lists, no parallelism: every lane walks them identically.
struct IiDev { const DevSide *a, *b; int ka, kb, step, on; long long agap, bgap; double spb; };
__device__ __forceinline__ long long ii_pos(const DevSide &s, int k) { return k < s.npfq ? (long long) s.pfq_pos[k] : (long long) s.len * s.pfq_step; }
__device__ __forceinline__ bool ii_before(const DevSide &s, int k, int col) { return k < s.npfq && (long long) s.pfq_pos[k] < (long long) col * s.pfq_step; }
__device__ void ii_init(IiDev &I, const DevProb &P)
{
    I.a = &P.a; I.b = &P.b; I.ka = I.kb = 0; I.agap = I.bgap = 0; I.spb = P.spb_fact;
    I.on = P.spb_fact > 0 && (P.a.npfq > 0 || P.b.npfq > 0);
    if (!I.on) return;
    while (I.ka < P.a.npfq && (long long) P.a.pfq_pos[I.ka] < (long long) P.a.left * P.a.pfq_step) ++I.ka;
    while (I.kb < P.b.npfq && (long long) P.b.pfq_pos[I.kb] < (long long) P.b.left * P.b.pfq_step) ++I.kb;
    I.step = P.a.npfq > 0 ? P.a.pfq_step : P.b.pfq_step;
    const int igap = P.a.left - P.b.left;
    if (igap > 0) I.bgap = (long long) igap * I.step; else I.agap = -(long long) igap * I.step;
}
__device__ double ii_store(IiDev &I, int m, int n)
{
    double scr = 0;
    bool an = ii_before(*I.a, I.ka, m), bn = ii_before(*I.b, I.kb, n);
    while (an || bn) {
        const long long apos = ii_pos(*I.a, I.ka) + I.agap, bpos = ii_pos(*I.b, I.kb) + I.bgap;
        if (an && bn && apos == bpos) scr += I.a->pfq_dns[I.ka] * I.b->pfq_dns[I.kb];
        if (an && apos <= bpos) { ++I.ka; an = ii_before(*I.a, I.ka, m); }
        if (bn && bpos <= apos) { ++I.kb; bn = ii_before(*I.b, I.kb, n); }
    }
    return I.spb * scr;
}
#define SP_FAST_LIST 192
template <int KIND, bool STREAM>
__device__ void sp_calcskl(const DevProb &P, const SpParamsDev &sp, const int2 *skl, int nskl, double *out, int *gepws, int2 *fast_lists, const SpStage &G, SpStream &S)
{
    // Gep1st of both sides (fspscore.h:146-147: alprm.ls > 2); the workspace arrives zeroed
    const bool gep = KIND >= 1 && P.noll == 3 && gepws != 0;
    GepDev agep, bgep;
    NtvState N;
    N.unit = 0; N.gla = N.glb = 0;
    if (KIND == 3) {                                           // workspace: gla[an], glb[bn], then the rings (Noll 3)
        N.gla = gepws; N.glb = gepws + P.a.many;
        gepws += P.a.many + P.b.many + 2;
        N.unit = (P.a.weight && P.b.weight) ? (P.a.many == 1 ? 1 : P.b.many == 1 ? 3 : 2) : 0;   // PreSpScore::calcSpScore, fspscore.cc:598-610
        if (N.unit == 3) sp_pregap(P.a, N.gla);                // (glb = 0, fspscore.h:58-66)
        else if (N.unit != 1) { sp_pregap(P.a, N.gla); sp_pregap(P.b, N.glb); }
    }
    agep.many = P.a.many; bgep.many = P.b.many; agep.k1 = bgep.k1 = P.codonk1;
    agep.w = P.a.weight; bgep.w = P.b.weight;
    agep.q = gepws; agep.qp = gepws + (size_t) P.a.many * P.codonk1;
    bgep.q = agep.qp + P.a.many; bgep.qp = bgep.q + (size_t) P.b.many * P.codonk1;
    double lunp = 0;
    DList dla, dlb;
    dla.p = P.dla[XH]; dla.s = P.spw; dlb.p = P.dlb[XH]; dlb.s = P.spw;
    // the two running lists are touched several times per column by a dependent chain: in LDS (the kernel's 2 x SP_FAST_LIST
    // entries) when they fit -- a global round trip per access made a _pf alignment cost 17 us per column
    if (fast_lists && P.capa + 1 <= SP_FAST_LIST && P.capb + 1 <= SP_FAST_LIST) { dla.p = fast_lists; dla.s = 1; dlb.p = fast_lists + SP_FAST_LIST; dlb.s = 1; }
    if (KIND == 1 || KIND == 2) cleardelta(dla);
    if (KIND == 2) cleardelta(dlb);
    int m = skl[0].x, n = skl[0].y, glb = 0;
    int apos = m - 1, bpos = n - 1;
    double scr = 0, tgap = 0;
    SpStat St; St.mch = St.mmc = St.unp = 0;
    IiDev II;
    ii_init(II, P);
    int col = 0;                                               // path columns walked so far = index of the next slot of the stream
    if (STREAM) sps_fetch(S, 0);
    for (int k = 1; k < nskl; ++k) {
        const int mi = skl[k].x - m, ni = skl[k].y - n, i = mi - ni;
        auto run = [&](int mi_, int ni_) {
            if (KIND == 3) sp_calscr_ntv(P, N, mi_, ni_, apos, bpos, scr, tgap, gep, agep, bgep, lunp, St);
            else sp_calscr<(KIND == 3 ? 0 : KIND), STREAM>(P, sp, mi_, ni_, apos, bpos, glb, dla, dlb, scr, tgap, gep, agep, bgep, lunp, St, G, S, col);
        };
        if (!i || !mi || !ni) run(mi, ni);
        else if (i > 0) { run(ni, ni); run(i, 0); }
        else { run(mi, mi); run(0, -i); }
        if (II.on) {                                               // fspscore.h:231-247
            int d = i >= 0 ? ni : mi;
            if (d) { m += d; n += d; scr += ii_store(II, m, n); }
            if (i < 0) { d = -i; n -= i; } else if (i > 0) { d = i; m += i; } else d = 0;
            if (d) {
                if (i > 0) II.bgap += (long long) d * II.step; else II.agap += (long long) d * II.step;
                scr += ii_store(II, m, n);
            }
        }
        m = skl[k].x; n = skl[k].y;
    }
    gep_wave_sync();
    scr += tgap * (KIND == 1 ? P.weighted_gop : P.basic_gop) + sp.diff_u * lunp;    // wgop(tgap, lunp), maln.h:321-325
    out[0] = scr / sp.vab;                                                       // rescale
    out[1] = tgap / sp.vab;
    out[2] = scr;
    out[3] = St.mch / sp.vab; out[4] = St.mmc / sp.vab; out[5] = St.unp / sp.vab;     // PwdM::rescale, maln2.cc:249-250
}
// ---- the streamed walk with the lists ACROSS the lanes -----------------------------------------------------------------------
// What is carried from column to column are the two running lists; the walker above keeps them in LDS and every lane replays
// the same scalar merges, entry after entry, each entry a dependent LDS read.  Here lane j holds entry j of a running list
// ({glen, nins}; the terminator {INT_MAX, 0} and everything behind it likewise) and lane i entry i of a column's static list, and
// the merges of gfreq.cc become a few uniform steps over the lanes:
//   GapLenSD(g, D)      = g + nins of the last entry of D whose glen <= g: a loop over D's entries (2-4), v_readlane each;
//   newgap(cf, Dc, df, Dd): both stretched sequences ascend strictly (glen ascends, nins ascends: newdelta only emits larger
//                         ones, incdelta adds one to all), so the reference's trailing cursor into cf is, for entry d of df,
//                         the NUMBER of cf entries stretched shorter than it; the products are then added in d order;
//   newdelta(D, df)     : the new list is {0, 0}, then {g + 1, nins} of every df entry whose looked-up nins exceeds its
//                         predecessor's, then the terminator -- a ballot and one pass over its set bits.
// Every floating-point operation is the reference's, in its order.  Lists of more than 62 entries, Noll 3 (the Gep1st rings read
// the lists through memory) and unstreamed alignments stay with the walker above; so does an alignment this one gives up on.
struct LaneList { int g, n, cnt; };                               // cnt (uniform): entries in front of the terminator
struct LaneStat { int g; double f; int cnt; };                    // a static list: cnt entries in front of its terminator
__device__ __forceinline__ int ll_rl(const int v, const int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double ll_rld(const double v, const int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void ll_clear(LaneList &D) { const int lane = threadIdx.x & 63; D.g = lane == 0 ? 0 : INT_MAX; D.n = 0; D.cnt = 1; }
__device__ __forceinline__ int ll_nins_at(const LaneList &D, const int g)
{
    int r = ll_rl(D.n, 0);
    for (int j = 1; j < D.cnt; ++j) { const int gj = ll_rl(D.g, j), nj = ll_rl(D.n, j); r = g >= gj ? nj : r; }
    return r;
}
__device__ __forceinline__ LaneStat ll_static(const SpSlot *sl, const int l, const DevSide &sd, const int view, const int pos, bool &ok)
{
    const int lane = threadIdx.x & 63;
    LaneStat L;
    int len = sl->len[l];
    if (len >= 0) {
        L.g = lane < len ? sl->glen[l][lane & (SPS_W - 1)] : -1;
        L.f = lane < len ? sl->freq[l][lane & (SPS_W - 1)] : 0;
    } else {                                                   // longer than a slot holds: read in place
        const int o = sd.off[view][pos + 1];
        len = sd.off[view][pos + 2] - o;
        L.g = lane < len ? sd.glen[view][o + lane] : -1;
        L.f = lane < len ? sd.freq[view][o + lane] : 0;
    }
    const unsigned long long neg = __ballot(L.g < 0);          // the terminator ends the list, wherever the count says it is
    L.cnt = neg ? (int) __ffsll((long long) neg) - 1 : 64;
    if (L.cnt >= 63) ok = false;
    return L;
}
__device__ __forceinline__ double ll_newgap4(const LaneStat &cf, const LaneList &Dc, const LaneStat &df, const LaneList &Dd)
{   // newgap(cf, dlc, df, dld), gfreq.cc:507-521
    if (cf.cnt == 0 || df.cnt == 0) return 0;
    const int ic = cf.g + ll_nins_at(Dc, cf.g), jd = df.g + ll_nins_at(Dd, df.g);
    int cur = 0;                                               // lane d: the cursor into cf when df's entry d is reached
    double cfq = ll_rld(cf.f, 0);
    for (int c = 0; c < cf.cnt; ++c) {
        const bool shorter = ll_rl(ic, c) < jd;
        cur += shorter ? 1 : 0;
        const double nxt = c + 1 < cf.cnt ? ll_rld(cf.f, c + 1) : 0;
        cfq = shorter ? nxt : cfq;                             // = cf.freq[cur] while cur < cf.cnt
    }
    const double term = cfq * df.f;
    double g = 0;
    for (int d = 0; d < df.cnt; ++d) {
        if (ll_rl(cur, d) >= cf.cnt) break;
        g += ll_rld(term, d);
    }
    return g;
}
__device__ __forceinline__ double ll_newgap_cj(const LaneStat &cf, const LaneList &Dc, const int j)
{   // newgap(cf, dlc, j), gfreq.cc:523-532
    const int lane = threadIdx.x & 63;
    const int ic = cf.g + ll_nins_at(Dc, cf.g);
    const unsigned long long hit = __ballot(lane < cf.cnt && ic >= j);
    return hit ? ll_rld(cf.f, (int) __ffsll((long long) hit) - 1) : 0;
}
__device__ __forceinline__ double ll_newgap_di(const LaneStat &df, const int i, const LaneList &Dd)
{   // newgap(df, i, dld), gfreq.cc:534-545
    const int lane = threadIdx.x & 63;
    const int jd = df.g + ll_nins_at(Dd, df.g);
    const unsigned long long stop = __ballot(lane < df.cnt && i < jd);
    const int lim = stop ? (int) __ffsll((long long) stop) - 1 : df.cnt;
    double g = 0;
    for (int d = 0; d < lim; ++d) g += ll_rld(df.f, d);
    return g;
}
__device__ __forceinline__ void ll_newdelta(LaneList &D, const LaneStat &df, bool &ok)
{   // newdelta(dlt, df, dln, 1), gfreq.cc:570-587, in place
    const int lane = threadIdx.x & 63;
    const int sn = lane < df.cnt ? ll_nins_at(D, df.g) : 0;
    const int up = __builtin_amdgcn_update_dpp(0, sn, 0x138, 0xf, 0xf, false);      // wave_shr:1 -- lane i gets lane i - 1's, lane 0 gets 0
    unsigned long long m = __ballot(lane < df.cnt && sn > (lane == 0 ? 0 : up));
    int ng = lane == 0 ? 0 : INT_MAX, nn = 0, p = 1;
    while (m) {
        const int src = (int) __ffsll((long long) m) - 1;
        m &= m - 1;
        const int gg = ll_rl(df.g, src) + 1, sv = ll_rl(sn, src);
        ng = lane == p ? gg : ng;
        nn = lane == p ? sv : nn;
        ++p;
    }
    if (p >= 63) ok = false;
    D.g = ng; D.n = nn; D.cnt = p;
}
__device__ __forceinline__ void ll_incdelta(LaneList &D)
{   // incdelta(dlt, dln, 1), gfreq.cc:598-605, in place
    const int lane = threadIdx.x & 63;
    D.n += lane < D.cnt ? 1 : 0;
}
template <int KIND>
__device__ bool sp_calcskl_lanes(const DevProb &P, const SpParamsDev &sp, const int2 *skl, const int nskl, double *out, SpStream &S)
{
    const DevSide &a = P.a, &b = P.b;
    bool ok = true;
    LaneList Da, Db;
    ll_clear(Da); ll_clear(Db);
    int m = skl[0].x, n = skl[0].y, glb = 0;
    int apos = m - 1, bpos = n - 1;
    double scr = 0, tgap = 0;
    SpStat St; St.mch = St.mmc = St.unp = 0;
    IiDev II;
    ii_init(II, P);
    int col = 0;
    sps_fetch(S, 0);
    const bool stats = !(sp.flags & 1);
    auto run = [&](int mi, int ni) {
        if (mi == ni) {
            while (mi-- && ok) {
                ++apos; ++bpos;
                const SpSlot *sl = sps_col(S, col); ++col;
                scr += sl->cs;
                if (KIND == 1) {
                    const LaneStat at = ll_static(sl, 1, a, 1, apos, ok);
                    tgap += ll_newgap_di(at, glb, Da);
                    if (stats) sp_stt2(P, apos, bpos, true, true, St);
                    ll_newdelta(Da, at, ok);
                    glb = 0;
                } else {
                    const LaneStat as = ll_static(sl, 0, a, 0, apos, ok), at = ll_static(sl, 1, a, 1, apos, ok);
                    const LaneStat bs = ll_static(sl, 2, b, 0, bpos, ok), bt = ll_static(sl, 3, b, 1, bpos, ok);
                    tgap += ll_newgap4(as, Da, bt, Db)
                          + ll_newgap4(bs, Db, at, Da);
                    if (stats) sp_stt2(P, apos, bpos, true, true, St);
                    ll_newdelta(Da, at, ok);
                    ll_newdelta(Db, bt, ok);
                }
            }
        } else if (mi) {
            while (mi-- && ok) {
                ++apos;
                const SpSlot *sl = sps_col(S, col); ++col;
                const LaneStat as = ll_static(sl, 0, a, 0, apos, ok), at = ll_static(sl, 1, a, 1, apos, ok);
                scr += sl->cs;
                if (KIND == 1) {
                    tgap += ll_newgap_cj(as, Da, glb);
                    if (stats) sp_stt2(P, apos, bpos, true, false, St);
                    ll_newdelta(Da, at, ok);
                    ++glb;
                } else {
                    const LaneStat br = ll_static(sl, 4, b, 2, bpos, ok);
                    tgap += ll_newgap4(as, Da, br, Db);
                    if (stats) sp_stt2(P, apos, bpos, true, false, St);
                    ll_newdelta(Da, at, ok);
                    ll_incdelta(Db);
                }
            }
        } else if (ni) {
            while (ni-- && ok) {
                ++bpos;
                const SpSlot *sl = sps_col(S, col); ++col;
                scr += sl->cs;
                if (KIND == 1) {
                    const LaneStat ar = ll_static(sl, 2, a, 2, apos, ok);
                    tgap += ll_newgap_di(ar, glb, Da);
                    if (stats) sp_stt2(P, apos, bpos, false, true, St);
                    ll_incdelta(Da);
                } else {
                    const LaneStat bs = ll_static(sl, 2, b, 0, bpos, ok), bt = ll_static(sl, 3, b, 1, bpos, ok);
                    const LaneStat ar = ll_static(sl, 4, a, 2, apos, ok);
                    tgap += ll_newgap4(bs, Db, ar, Da);
                    if (stats) sp_stt2(P, apos, bpos, false, true, St);
                    ll_newdelta(Db, bt, ok);
                    ll_incdelta(Da);
                }
            }
        }
    };
    for (int k = 1; k < nskl && ok; ++k) {
        const int mi = skl[k].x - m, ni = skl[k].y - n, i = mi - ni;
        if (!i || !mi || !ni) run(mi, ni);
        else if (i > 0) { run(ni, ni); run(i, 0); }
        else { run(mi, mi); run(0, -i); }
        if (II.on) {                                               // fspscore.h:231-247
            int d = i >= 0 ? ni : mi;
            if (d) { m += d; n += d; scr += ii_store(II, m, n); }
            if (i < 0) { d = -i; n -= i; } else if (i > 0) { d = i; m += i; } else d = 0;
            if (d) {
                if (i > 0) II.bgap += (long long) d * II.step; else II.agap += (long long) d * II.step;
                scr += ii_store(II, m, n);
            }
        }
        m = skl[k].x; n = skl[k].y;
    }
    if (!ok) return false;
    scr += tgap * (KIND == 1 ? P.weighted_gop : P.basic_gop) + sp.diff_u * 0.;      // wgop(tgap, lunp), maln.h:321-325 (lunp = 0 without Gep1st)
    out[0] = scr / sp.vab;
    out[1] = tgap / sp.vab;
    out[2] = scr;
    out[3] = St.mch / sp.vab; out[4] = St.mmc / sp.vab; out[5] = St.unp / sp.vab;
    return true;
}
// The pre-pass of the streamed walk: one thread per path column of every streamed alignment.  colpre[k] = path columns of the
// skeleton's segments 1..k (host: a segment from corner k-1 to corner k has max(rows, columns) of them, its diagonal part first).
__device__ __forceinline__ void sps_pack(SpSlot &o, const int l, const DevSide &sd, const int view, const int pos)
{
    const int off = sd.off[view][pos + 1], len = sd.off[view][pos + 2] - off;       // entries incl. the terminator
    if (len > SPS_W) { o.len[l] = -1; return; }
    o.len[l] = len;
    for (int k = 0; k < len; ++k) { o.glen[l][k] = sd.glen[view][off + k]; o.freq[l][k] = sd.freq[view][off + k]; }
}
#ifdef G2G_TU_V1
extern "C" __global__ void __launch_bounds__(256)
g2g_spprep_kernel(const DevProb *probs, int nprob, int nreal, const int2 *skl, const int *skl_off, const int *nskl, const int *colpre,
                  const long long *slot_off, SpSlot *slots)
{
    const int ip = blockIdx.y;                               // entry ip walks problem ip % nreal (g2g_batch_spscore_sets)
    if (ip >= nprob || slot_off[ip] < 0) return;
    const DevProb &P = probs[ip % nreal];
    const int2 *s = skl + skl_off[ip];
    const int *cp = colpre + skl_off[ip];
    const int ns = nskl[ip], ncols = cp[ns - 1];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) {
        int lo = 1, hi = ns - 1;                               // the segment of column c: the first k with colpre[k] > c
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cp[mid] > c) hi = mid; else lo = mid + 1; }
        const int k = lo, t = c - cp[k - 1];
        const int m0 = s[k - 1].x, n0 = s[k - 1].y, mi = s[k].x - m0, ni = s[k].y - n0;
        const int d = mi < ni ? mi : ni;                       // the diagonal part comes first (fspscore.h:222-230)
        int apos, bpos, type;                                  // type: 0 both advance, 1 a alone, 2 b alone
        if (t < d) { type = 0; apos = m0 + t; bpos = n0 + t; }
        else if (mi > ni) { type = 1; apos = m0 + t; bpos = n0 + d - 1; }
        else { type = 2; bpos = n0 + t; apos = m0 + d - 1; }
        SpSlot &o = slots[slot_off[ip] + c];
        for (int l = 0; l < 5; ++l) o.len[l] = 0;
        o.pad = 0;
        o.cs = type == 0 ? sim2(P, apos, bpos) : type == 1 ? unpa(P, apos, bpos) : unpb(P, bpos, apos);
        if (P.kind == 1) {
            if (type == 0) sps_pack(o, 1, P.a, 1, apos);
            else if (type == 1) { sps_pack(o, 0, P.a, 0, apos); sps_pack(o, 1, P.a, 1, apos); }
            else sps_pack(o, 2, P.a, 2, apos);
        } else {
            if (type != 2) { sps_pack(o, 0, P.a, 0, apos); sps_pack(o, 1, P.a, 1, apos); }
            if (type != 1) { sps_pack(o, 2, P.b, 0, bpos); sps_pack(o, 3, P.b, 1, bpos); }
            if (type == 1) sps_pack(o, 4, P.b, 2, bpos);
            if (type == 2) sps_pack(o, 4, P.a, 2, apos);
        }
    }
}
extern "C" __global__ void __launch_bounds__(64)
g2g_spscore_kernel(const DevProb *probs, int nprob, int nreal, const SpParamsDev *sp, const int2 *skl, const int *skl_off, const int *nskl,
                   double *out, int *status, int *gepws, const long long *gep_off, const int *colpre, const long long *slot_off, const SpSlot *slots, int nolanes)
{
    const int ip = blockIdx.x;
    if (ip >= nprob) return;                                   // (all 64 lanes walk the chain in lockstep, see GepDev)
    const DevProb &P = probs[ip % nreal];
    for (int k = 0; k < 6; ++k) out[6 * ip + k] = 0;
    if (P.kind < 0) { status[ip] = -1; return; }
    int *ws = (gepws && gep_off[ip] >= 0) ? gepws + gep_off[ip] : (int *) 0;
    if (((P.noll == 3 && P.kind >= 1) || P.kind == 3) && !ws) { status[ip] = -2; return; }
    if (P.kind == 3 && (!P.a.gapdens || !P.b.gapdens)) { status[ip] = -1; return; }
    if (nskl[ip] < 2) { status[ip] = -1; return; }
    const int2 *s = skl + skl_off[ip];
    __shared__ int2 sp_lists[2 * SP_FAST_LIST];
    __shared__ int sp_sgl[6 * SP_STAGE];
    __shared__ double sp_sfq[6 * SP_STAGE];
    __shared__ SpSlot sp_chunk[SPS_CH];
    int2 *fl = (int2 *) sp_lists;
    SpStage G; G.gl = (int *) sp_sgl; G.fq = (double *) sp_sfq;
    SpStream S;
    const bool stream = slots && slot_off[ip] >= 0 && (P.kind == 1 || P.kind == 2);
    S.g = stream ? slots + slot_off[ip] : (const SpSlot *) 0; S.lds = (SpSlot *) sp_chunk; S.ncols = stream ? colpre[skl_off[ip] + nskl[ip] - 1] : 0;
    if (P.kind == 0) sp_calcskl<0, false>(P, sp[ip], s, nskl[ip], out + 6 * ip, ws, fl, G, S);
    else if (P.kind == 1) {
        const bool lanes = stream && !(P.noll == 3 && ws) && !nolanes;
        if (lanes && sp_calcskl_lanes<1>(P, sp[ip], s, nskl[ip], out + 6 * ip, S)) { }
        else if (stream) sp_calcskl<1, true>(P, sp[ip], s, nskl[ip], out + 6 * ip, ws, fl, G, S);
        else sp_calcskl<1, false>(P, sp[ip], s, nskl[ip], out + 6 * ip, ws, fl, G, S);
    } else if (P.kind == 2) {
        const bool lanes = stream && !(P.noll == 3 && ws) && !nolanes;
        if (lanes && sp_calcskl_lanes<2>(P, sp[ip], s, nskl[ip], out + 6 * ip, S)) { }
        else if (stream) sp_calcskl<2, true>(P, sp[ip], s, nskl[ip], out + 6 * ip, ws, fl, G, S);
        else sp_calcskl<2, false>(P, sp[ip], s, nskl[ip], out + 6 * ip, ws, fl, G, S);
    }
    else sp_calcskl<3, false>(P, sp[ip], s, nskl[ip], out + 6 * ip, ws, fl, G, S);
    status[ip] = 0;
}
#else
extern "C" __global__ void g2g_spprep_kernel(const DevProb *probs, int nprob, int nreal, const int2 *skl, const int *skl_off, const int *nskl, const int *colpre,
                                             const long long *slot_off, SpSlot *slots);
extern "C" __global__ void g2g_spscore_kernel(const DevProb *probs, int nprob, int nreal, const SpParamsDev *sp, const int2 *skl, const int *skl_off, const int *nskl,
                                              double *out, int *status, int *gepws, const long long *gep_off, const int *colpre, const long long *slot_off, const SpSlot *slots, int nolanes);
#endif
