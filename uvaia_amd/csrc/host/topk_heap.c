/* topk_heap.c -- implementation of min_heap.h.  Behaviour (including the array layout after every operation) follows
 * src/min_heap.c:41-158 of the reference; written iteratively. */
#include "min_heap.h"

#include <stdlib.h>
#include <string.h>

#define N_KEYS 6

int
compare_q_item_score (int *a, int *b)
{
  for (int i = 0; i < N_KEYS; i++) if (a[i] != b[i]) return b[i] - a[i];
  return 0;
}

static inline int ahead (const q_item *a, const q_item *b) { return compare_q_item_score ((int *) a->score, (int *) b->score) < 0; }

heap_t
new_heap_t (int heap_size)
{
  heap_t pq = (heap_t) biomcmc_malloc (sizeof (struct heap_struct));
  pq->n = 0;
  pq->max_incompatible = 0xffffff;
  pq->heap_size = heap_size < 2 ? 2 : heap_size;
  pq->seq = (q_item *) biomcmc_malloc (((size_t) pq->heap_size + 1) * sizeof (q_item));
  memset (pq->seq, 0, ((size_t) pq->heap_size + 1) * sizeof (q_item));
  return pq;
}

void
del_heap_t (heap_t pq)
{
  if (!pq) return;
  if (pq->seq) { for (int i = 0; i <= pq->heap_size; i++) free (pq->seq[i].name); free (pq->seq); }
  free (pq);
}

static void
swap_items (q_item *a, q_item *b) { q_item t = *a; *a = *b; *b = t; }

static void
sink (heap_t pq, int p)
{ /* move slot p down while it ranks ahead of its worse child */
  for (;;) {
    int worst = p;
    for (int c = 2 * p; c <= 2 * p + 1 && c <= pq->n; c++) if (ahead (&pq->seq[worst], &pq->seq[c])) worst = c;
    if (worst == p) return;
    swap_items (&pq->seq[p], &pq->seq[worst]);
    p = worst;
  }
}

static void
swim (heap_t pq, int i)
{ /* move slot i up while its parent ranks ahead of it */
  for (; i > 1 && ahead (&pq->seq[i / 2], &pq->seq[i]); i /= 2) swap_items (&pq->seq[i / 2], &pq->seq[i]);
}

q_item heap_get_worse (heap_t pq) { return pq->seq[1]; }

q_item
heap_remove_worse (heap_t pq)
{
  q_item none; memset (&none, 0, sizeof none);
  if (!pq->n) return none;
  q_item top = pq->seq[1];
  pq->seq[1] = pq->seq[pq->n];
  memset (&pq->seq[pq->n], 0, sizeof (q_item));
  pq->n--;
  sink (pq, 1);
  return top;       /* ownership of top.name passes to the caller */
}

bool
heap_insert (heap_t pq, q_item item)
{
  int slot;
  if (pq->n == pq->heap_size) {
    if (!ahead (&item, &pq->seq[1])) return false;
    slot = 1;
  } else slot = ++pq->n;
  free (pq->seq[slot].name);
  pq->seq[slot].name = item.name ? strdup (item.name) : NULL;
  memcpy (pq->seq[slot].score, item.score, sizeof item.score);
  if (slot == 1) sink (pq, 1); else swim (pq, slot);
  return true;
}

static void
stable_sort_items (q_item *v, int n)
{ /* bottom-up merge sort: ties keep their order, which is what glibc's qsort (a merge sort) gives the reference */
  if (n < 2) return;
  q_item *tmp = (q_item *) biomcmc_malloc ((size_t) n * sizeof (q_item)), *src = v, *dst = tmp;
  for (int width = 1; width < n; width *= 2) {
    for (int lo = 0; lo < n; lo += 2 * width) {
      int mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n, a = lo, b = mid, o = lo;
      while (a < mid && b < hi) dst[o++] = ahead (&src[b], &src[a]) ? src[b++] : src[a++];
      while (a < mid) dst[o++] = src[a++];
      while (b < hi) dst[o++] = src[b++];
    }
    q_item *t = src; src = dst; dst = t;
  }
  if (src != v) memcpy (v, src, (size_t) n * sizeof (q_item));
  free (tmp);
}

void
heap_finalise_heap_qsort (heap_t pq)
{
  swap_items (&pq->seq[0], &pq->seq[pq->n]);       /* [1..n] -> [0..n-1] */
  stable_sort_items (pq->seq, pq->n);
  if (pq->n < pq->heap_size - 1) {
    for (int i = pq->n; i <= pq->heap_size; i++) free (pq->seq[i].name);
    pq->seq = (q_item *) biomcmc_realloc (pq->seq, ((size_t) pq->n + 1) * sizeof (q_item));
    memset (&pq->seq[pq->n], 0, sizeof (q_item));   /* keeps del_heap_t's "slots 0..heap_size" invariant */
    pq->heap_size = pq->n;
  }
}
