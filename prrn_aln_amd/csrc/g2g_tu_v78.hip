// g2g_tu_v78.hip -- one translation unit of libg2g.so (g2g_device.h): emits the G2G_TU_V78 kernel group, sees the device functions of the others
#define G2G_TU_V78 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/g2g.h"
#include "g2g_device.h"
#include "g2g_internal.h"
#include "g2g_kernels.hip"
#include "g2g_kernels_v2.hip"
#include "g2g_kernels_v3.hip"
#include "g2g_kernels_v6.hip"
#include "g2g_kernels_v7.hip"
#include "g2g_kernels_v8.hip"
