/* oracle/pbd_oracle.c -- TEST INFRASTRUCTURE ONLY: plain-C CPU restatement of the reference's
 * hot path (see pbd_oracle_impl.h).  Built by oracle/Makefile into oracle/_ref/libpbd_oracle.so
 * with -O2 -ffp-contract=off.  Two instantiations: po32_* (float build of the reference) and
 * po64_* (double build).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it. */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pbd_oracle.h"

#define PO_CAT2(a, b) a##b
#define PO_CAT(a, b) PO_CAT2(a, b)

#define PO_REAL float
#define PO(name) PO_CAT(po32_, name)
#define PO_SQRT sqrtf
#define PO_FABS fabsf
#define PO_ACOS acosf
#define PO_REAL_MAX FLT_MAX
#include "pbd_oracle_impl.h"
#undef PO_REAL
#undef PO
#undef PO_SQRT
#undef PO_FABS
#undef PO_ACOS
#undef PO_REAL_MAX

#define PO_REAL double
#define PO(name) PO_CAT(po64_, name)
#define PO_SQRT sqrt
#define PO_FABS fabs
#define PO_ACOS acos
#define PO_REAL_MAX DBL_MAX
#include "pbd_oracle_impl.h"
