// g2g_host.cpp -- host-side pieces of the operator (no device code): stdskl.
// (The level-1 builders -- mode selection, thickness, profile vectors, gap profiles -- land here too.)
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "../../include/g2g.h"
#include "g2g_internal.h"

// <-> stdskl(SKL**), reference src/gaps.cc:139-174: order the raw traceback by (m, n), drop repeats and
// inconsistent steps, and make every diagonal/gap junction explicit.  Output: ascending corners
// (the reference's skl[1..n]; its skl[0] header and EOS sentinel are left to the caller's wrapper).
extern "C" g2g_skl *g2g_stdskl(const g2g_skl *in, int num, int *nout)
{
    if (!nout || num < 0 || (num && !in)) return NULL;
    if (num < 2) {
        g2g_skl *one = (g2g_skl *) malloc(sizeof(g2g_skl) * (num > 0 ? num : 1));
        if (num) one[0] = in[0];
        *nout = num;
        return one;
    }
    g2g_skl *org = (g2g_skl *) malloc(sizeof(g2g_skl) * num);
    memcpy(org, in, sizeof(g2g_skl) * num);
    // scmpf (gaps.cc:116-121) is a total order on (m, n): any sort gives the reference's sequence
    std::sort(org, org + num, [](const g2g_skl &x, const g2g_skl &y) { return x.m != y.m ? x.m < y.m : x.n < y.n; });
    g2g_skl *std_ = (g2g_skl *) malloc(sizeof(g2g_skl) * (2 * (size_t) num + 1));
    int w = 0, pr = 2;
    const g2g_skl *prv = org;
    for (int i = 1; i < num; ++i) {
        const g2g_skl *cur = org + i;
        const int dm = cur->m - prv->m, dn = cur->n - prv->n;
        if (!dm && !dn) continue;                 // no increment
        if (dm < 0 || dn < 0) continue;           // inconsistent
        const int dd = std::min(dm, dn);
        int df = dn - dm;
        if (df) df = df > 0 ? 1 : -1;
        if (dd && df) {                           // diagonal run followed by a gap: interpolate the junction
            if (pr) std_[w++] = *prv;
            std_[w].m = prv->m + dd; std_[w].n = prv->n + dd; ++w;
        } else if (df != pr || !dm) std_[w++] = *prv;
        pr = df;
        prv = cur;
    }
    std_[w++] = *prv;
    free(org);
    *nout = w;
    return std_;
}
