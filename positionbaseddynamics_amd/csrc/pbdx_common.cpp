// pbdx_common.cpp -- error reporting and the constraint-type table.
#include "pbdx_internal.h"
#include <string.h>

namespace pbdx {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
}
void clear_error() { g_error[0] = 0; }

// algorithmic bytes per projection: SURVEY.md section 8d (fp32, 4-byte indices,
// every endpoint position read once and written once, no cache credit)
static const TypeInfo kTypes[PBDX_NUM_CONSTRAINT_TYPES] = {
	{ "DistanceConstraint",               2, 2,  68,  false },
	{ "DistanceConstraint_XPBD",          2, 2,  76,  true  },
	{ "DihedralConstraint",               4, 2,  132, false },
	{ "IsometricBendingConstraint",       4, 17, 192, false },
	{ "IsometricBendingConstraint_XPBD",  4, 17, 200, true  },
	{ "FEMTriangleConstraint",            3, 10, 116, false },
	{ "StrainTriangleConstraint",         3, 9,  112, false },
	{ "VolumeConstraint",                 4, 2,  132, false },
	{ "VolumeConstraint_XPBD",            4, 2,  140, true  },
	{ "FEMTetConstraint",                 4, 12, 168, false },
	{ "XPBD_FEMTetConstraint",            4, 12, 176, true  },
	{ "StrainTetConstraint",              4, 13, 164, false },
	{ "ShapeMatchingConstraint",          4, 24, 204, false },
};

const TypeInfo *type_info(int type)
{
	if (type < 0 || type >= PBDX_NUM_CONSTRAINT_TYPES) return nullptr;
	return &kTypes[type];
}

} // namespace pbdx

extern "C" {

const char *pbdx_last_error(void) { return pbdx::g_error; }
int pbdx_version(void) { return PBDX_VERSION; }
uint32_t pbdx_type_num_bodies(int type) { const pbdx::TypeInfo *t = pbdx::type_info(type); return t ? t->num_bodies : 0; }
uint32_t pbdx_type_param_stride(int type) { const pbdx::TypeInfo *t = pbdx::type_info(type); return t ? t->param_stride : 0; }
uint32_t pbdx_type_algorithmic_bytes(int type) { const pbdx::TypeInfo *t = pbdx::type_info(type); return t ? t->algorithmic_bytes : 0; }
const char *pbdx_type_name(int type) { const pbdx::TypeInfo *t = pbdx::type_info(type); return t ? t->name : ""; }

// contiguous block [begin, end) of `total` independent instances owned by `rank` of `world` (sizes differ by at most one)
int pbdx_ensemble_shard(uint64_t total, uint32_t world, uint32_t rank, uint64_t *begin, uint64_t *end)
{
	if (world == 0 || rank >= world || !begin || !end) { pbdx::set_error("pbdx_ensemble_shard: bad world / rank"); return PBDX_ERR_INVALID; }
	const uint64_t base = total / world, extra = total % world;
	*begin = (uint64_t)rank * base + (rank < extra ? rank : extra);
	*end = *begin + base + (rank < extra ? 1u : 0u);
	return PBDX_OK;
}

}
