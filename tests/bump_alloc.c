/* Test helper: the bump allocator of jpegxl-rs/src/memory.rs:50-105 (BumpManager) as a C callback pair, so that the
   allocation callbacks run without the Python interpreter in the way.  alloc never reuses, free only counts. */
#include <stdatomic.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { uint8_t* arena; size_t size; _Atomic size_t footer; _Atomic size_t allocs, frees, failed, largest; } Bump;

Bump* bump_create(size_t n) { Bump* b = calloc(1, sizeof(Bump)); b->arena = malloc(n); b->size = n; return b; }
void bump_destroy(Bump* b) { free(b->arena); free(b); }
void* bump_alloc(void* opaque, size_t size) {
  Bump* b = opaque;
  size_t rounded = (size + 63) & ~(size_t)63;      /* (the Rust test hands out unaligned addresses; libjxl over-aligns itself) */
  size_t at = atomic_fetch_add(&b->footer, rounded);
  atomic_fetch_add(&b->allocs, 1);
  if (size > b->largest) b->largest = size;
  if (at + rounded > b->size) { atomic_fetch_add(&b->failed, 1); return NULL; }
  return b->arena + at;
}
void bump_free(void* opaque, void* p) { (void)p; atomic_fetch_add(&((Bump*)opaque)->frees, 1); }
void bump_stats(Bump* b, size_t out[5]) { out[0] = b->footer; out[1] = b->allocs; out[2] = b->frees; out[3] = b->failed; out[4] = b->largest; }
