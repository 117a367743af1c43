/*
 * gpu_glue.h -- the few lines of host code that connect uvaia's own structures (query_t, heap_t) to the C ABI of
 * include/uvaia_gpu.h.  This is what a maintainer of the reference adds next to src/nearest.c.
 */
#ifndef UVAIA_HOST_GPU_GLUE_H
#define UVAIA_HOST_GPU_GLUE_H

#include "../../../include/uvaia_gpu.h"
#include "fastaseq.h"
#include "min_heap.h"

#ifdef __cplusplus
extern "C" {
#endif

/* uvaia_gpu_open() from a prepared query_t (after create_query_indices / reorder_query_structure) */
int uvaia_gpu_open_query (uvaia_gpu_ctx **ctx, query_t query, int heap_size, int device, size_t max_pool);
int uvaia_gpu_open_query_tuned (uvaia_gpu_ctx **ctx, query_t query, int heap_size, int device, size_t max_pool, const uvaia_gpu_tuning *tuning);

/* Fills n_query host heaps (made with new_heap_t(heap_size)) from the device heaps, slot for slot, including
 * max_incompatible; name_of(ordinal, user) must return the reference name for an ordinal (it is strdup()ed, as
 * heap_insert does, src/min_heap.c:101,112).  Afterwards heap_finalise_heap_qsort() gives the output order. */
int uvaia_gpu_collect_heaps (uvaia_gpu_ctx *ctx, heap_t *heap, const char *(*name_of) (int64_t ordinal, void *user), void *user);

/* the same two for several GPUs (uvaia_gpu_group_*, include/uvaia_gpu.h): devices[] = HIP devices of the members */
int uvaia_gpu_group_open_query (uvaia_gpu_group **group, query_t query, int heap_size, const int *devices, int n_devices, size_t max_pool, size_t piece_refs);
int uvaia_gpu_group_collect_heaps (uvaia_gpu_group *group, heap_t *heap, const char *(*name_of) (int64_t ordinal, void *user), void *user);
/* "0,2-4" -> {0,2,3,4}; returns the number of devices (0 = syntax error), at most max */
int uvaia_parse_device_list (const char *text, int *devices, int max);

#ifdef __cplusplus
}
#endif
#endif
