// g2g_tu_v2.hip -- one translation unit of libg2g.so (g2g_device.h): emits the G2G_TU_V2 kernel group, sees the device functions of the others
#define G2G_TU_V2 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/g2g.h"
#include "g2g_device.h"
#include "g2g_internal.h"
#include "g2g_kernels.hip"
#include "g2g_kernels_v2.hip"
#ifdef G2G_V2_STAMP
// diagnostics build (G2G_EXTRA_FLAGS=-DG2G_V2_STAMP): s_memtime per phase of the v2 step loop, summed over the first lane of every
// wave; [0..7] wave 0 of each workgroup, [8..15] the others.  Slots: 1 sources, 2 merges, 3 decisions, 4 list updates, 5 record
// scalars + trace byte, 6 parking / boundary stores / column ring, 7 the barrier at the end of the step
extern "C" void g2g_v2_stamps(unsigned long long *out, int reset)
{
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (out) (void) hipMemcpyFromSymbol(out, HIP_SYMBOL(g2g_stamp_acc), sizeof z);
    if (reset) (void) hipMemcpyToSymbol(HIP_SYMBOL(g2g_stamp_acc), z, sizeof z);
}
#endif
