// g2g_group.h -- the level-1 group object (<-> mSeq) as the translation units of libg2g.so share it (not installed):
// g2g_host.cpp builds its derived arrays on the host, g2g_build.hip on the device.
#ifndef G2G_GROUP_H_
#define G2G_GROUP_H_
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include <memory>
#include "../../include/g2g.h"

struct GapProfile {                                          // <-> class Gfq, src/gfreq.h:44-64
    int hetero;
    std::vector<int32_t> off[3], glen[3];
    std::vector<double> freq[3];
};


// <-> mSeq (src/mseq.h:86-200) reduced to what the group-to-group DP reads
struct g2g_group {
    int many, len, left, right;
    int molc, max_code;
    int dels, nils, exgl, exgr;
    bool has_weight;
    double tgapf;
    std::vector<uint8_t> seq;            // (len + 2) * many, position -1 first
    std::vector<double> weight;
    // derived lazily, like the reference (mkthick / convseq / Gfq on first use)
    bool thk_done;
    double sumwt;
    int thk_len;
    bool has_internalres;
    std::vector<int> internal_pos;       // position of mSeq::internalres[i]
    std::vector<double> thk;             // (thk_len + 2) * 3, index -1 first
    int vect, nelm, felm, simdim_used;
    std::vector<double> pseq;            // (len + 2) * nelm
    GapProfile *gfq;
    // device-resident twins of seq / weight / thk (only of groups with gaps: their rows ARE the per-position view) / pseq / the gap profile,
    // left behind by g2g_device_derive; the slabs they live in are shared by the groups of one batch and go back to the
    // context's pool with the last of them
    g2g_side_dev dev;
    std::vector<std::shared_ptr<void>> dev_slabs;
    // flattened views handed to the engine
    std::vector<double> thk_pos;         // (len + 2) * 3
    std::vector<double> gapdens, postgapdens;

    uint8_t at(int pos, int i) const { return seq[(size_t) (pos + 1) * many + i]; }
    uint8_t &at(int pos, int i) { return seq[(size_t) (pos + 1) * many + i]; }
    double *T(int j) { return &thk[(size_t) (j + 1) * 3]; }
    ~g2g_group() { delete gfq; }
};


// what a pairing asks of a group beyond its thickness: bit 0 the frequency vectors (VECTOR), bit 1 the profile vectors on top
// (VECPRO), bit 2 the gap profile
enum { G2G_NEED_VECTOR = 1, G2G_NEED_VECPRO = 2, G2G_NEED_GFQ = 4 };
struct g2g_ctx;
struct g2g_params;
// g2g_build.hip: thickness, vectors and gap profiles of n groups on the device, results in the
// groups' own arrays.  Returns G2G_OK, or an error the caller answers by building on the host (groups it does not take: nil
// codes, i.e. tgapf < 1 or local ends; more than 64 gap classes alive in one column).
int g2g_device_derive(g2g_ctx *ctx, const g2g_params *prm, int n, g2g_group *const *groups, const int *need);
#endif
