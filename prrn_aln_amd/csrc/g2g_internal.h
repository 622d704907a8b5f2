// g2g_internal.h -- declarations shared by the translation units of libg2g.so (not installed).
#ifndef G2G_INTERNAL_H_
#define G2G_INTERNAL_H_
#include "../../include/g2g.h"

void g2g_set_error(const char *fmt, const char *arg);

#endif
