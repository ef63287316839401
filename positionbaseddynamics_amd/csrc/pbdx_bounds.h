// pbdx_bounds.h -- the range checks of the sanitizer-grade debug build (-DPBDX_BOUNDS=1).
#ifndef PBDX_BOUNDS_H
#define PBDX_BOUNDS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// Sanitizer-grade debug build (scripts/build_variant.sh bounds -DPBDX_BOUNDS=1; tests/test_gpu_parity.py: test_bounds_checked_build_...).
// Every address of the fused / persistent sweep that is NOT behind a hardware-checked buffer descriptor -- particle ids read from the gid stream,
// the positions they select in HBM, LDS slots of the fill, the gather / scatter and the dictionary table, chunk and tile descriptor indices, the
// dependency lists of the persistent schedule -- is compared with the size of what it addresses before the access.  A violation is counted, the first one
// recorded (kind, workgroup, thread, index, limit, tile), and the access is skipped (stores) or redirected to element 0 (loads), so that a run completes
// and the host can read the record (pbdx_debug_bounds_report).  The product build compiles none of this (the macros fold to the plain access).
#ifndef PBDX_BOUNDS
#define PBDX_BOUNDS 0
#endif
enum { kBndGid = 1, kBndParticle = 2, kBndLdsFill = 3, kBndChunk = 4, kBndTable = 5, kBndLdsSlot = 6, kBndTile = 7, kBndDep = 8, kBndTableSrc = 9, kBndChunkRange = 10, kBndLdsIds = 11 };
#if PBDX_BOUNDS
static __device__ uint32_t g_bounds_rec[8];      // (one per translation unit; the checks all live in pbdx_sweep.hip, which also reads it out: sweep_bounds_report)
                                                 // [0] violations, first one: [1] kind, [2] workgroup, [3] thread, [4] index, [5] limit, [6] tile / aux
__device__ __forceinline__ bool bounds_ok(uint32_t kind, uint32_t index, uint32_t limit, uint32_t aux = 0u)
{
	if (index < limit) return true;
	if (atomicAdd(&g_bounds_rec[0], 1u) == 0u)
	{
		g_bounds_rec[1] = kind; g_bounds_rec[2] = blockIdx.x; g_bounds_rec[3] = threadIdx.x; g_bounds_rec[4] = index; g_bounds_rec[5] = limit; g_bounds_rec[6] = aux;
	}
	return false;
}
#define PBDX_BOK(kind, index, limit, aux) bounds_ok(kind, index, limit, aux)
#define PBDX_BCLAMP(kind, index, limit, aux) (bounds_ok(kind, index, limit, aux) ? (index) : 0u)
#else
#define PBDX_BOK(kind, index, limit, aux) true
#define PBDX_BCLAMP(kind, index, limit, aux) (index)
#endif

#endif
