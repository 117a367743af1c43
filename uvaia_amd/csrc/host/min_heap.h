/*
 * min_heap.h -- bounded "keep the k best" heap, host side.
 * Type and function names follow the reference's src/min_heap.h:14-39 so existing callers compile unchanged.  On the
 * GPU build the heaps live on the device while the search runs (uvaia_gpu.h); these host structures receive them at the
 * end (uvaia_gpu_collect_heaps, gpu_glue.h) and are also usable stand-alone.
 */
#ifndef UVAIA_HOST_MIN_HEAP_H
#define UVAIA_HOST_MIN_HEAP_H

#include "utils.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct heap_struct *heap_t;

typedef struct {
  int score[6];     /* compared lexicographically, larger is better */
  char *name;       /* owned copy */
} q_item;

struct heap_struct {
  q_item *seq;      /* slots 1..n hold a binary heap whose root (slot 1) is the WORST kept item */
  int heap_size, n;
  int max_incompatible;   /* mismatch tolerance derived from the root; not a heap property */
};

int compare_q_item_score (int *a, int *b);     /* <0 when a ranks ahead of b */
heap_t new_heap_t (int heap_size);             /* at least 2 slots */
void del_heap_t (heap_t pq);
q_item heap_get_worse (heap_t pq);
q_item heap_remove_worse (heap_t pq);
bool heap_insert (heap_t pq, q_item item);     /* copies the name; false if the item does not beat a full heap's root */
void heap_finalise_heap_qsort (heap_t pq);     /* items to slots 0..n-1, best first; ties keep their array order */

#ifdef __cplusplus
}
#endif
#endif
