// jxl-hip: host allocations through the caller's JxlMemoryManager (jpegxl-rs/src/memory.rs:24-39; jpegxl-sys memory_manager.rs).
// Every container of the parser / decoder is a `vec<T>` = std::vector with this allocator.  The manager in force is a per-thread
// scope set by the C ABI entry points of a decoder that was created with one (MmScope); each block remembers the `free` it must go
// back to in a 32-byte header, so a decoder may migrate between threads and scopes may nest.  Without a scope: malloc / free.
// Process-lifetime caches (natural coefficient orders, the DCT128/256 tables) deliberately stay on plain std::vector.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <vector>

namespace jxlhip {

struct MmHooks { void* opaque; void* (*alloc)(void*, size_t); void (*free)(void*, void*); };

inline const MmHooks*& MmCurrent() { static thread_local const MmHooks* cur = nullptr; return cur; }

struct MmScope {   // RAII: allocations made by this thread until destruction go through `h` (nullptr = malloc)
  const MmHooks* prev;
  explicit MmScope(const MmHooks* h) : prev(MmCurrent()) { MmCurrent() = h; }
  ~MmScope() { MmCurrent() = prev; }
};

struct MmHeader { void (*free)(void*, void*); void* opaque; uint64_t pad[2]; };   // 32 bytes: keeps 16-byte (and 32-byte) alignment of the payload
static_assert(sizeof(MmHeader) == 32, "header size");

inline void* MmAllocate(size_t bytes) {
  const MmHooks* h = MmCurrent();
  void* raw = h ? h->alloc(h->opaque, bytes + sizeof(MmHeader) + 32) : std::malloc(bytes + sizeof(MmHeader) + 32);
  if (!raw) throw std::bad_alloc();
  // the caller's allocator guarantees no alignment (memory_manager.rs:22-35): align the payload to 32 bytes ourselves
  uintptr_t p = (reinterpret_cast<uintptr_t>(raw) + sizeof(MmHeader) + 31) & ~(uintptr_t)31;
  MmHeader* hd = reinterpret_cast<MmHeader*>(p - sizeof(MmHeader));
  hd->free = h ? h->free : nullptr; hd->opaque = h ? h->opaque : nullptr;
  hd->pad[0] = reinterpret_cast<uintptr_t>(raw); hd->pad[1] = 0;
  return reinterpret_cast<void*>(p);
}
inline void MmDeallocate(void* ptr) {
  if (!ptr) return;
  MmHeader* hd = reinterpret_cast<MmHeader*>(reinterpret_cast<uintptr_t>(ptr) - sizeof(MmHeader));
  void* raw = reinterpret_cast<void*>((uintptr_t)hd->pad[0]);
  if (hd->free) hd->free(hd->opaque, raw); else std::free(raw);
}

template <class T> struct MmAlloc {
  typedef T value_type;
  MmAlloc() noexcept {}
  template <class U> MmAlloc(const MmAlloc<U>&) noexcept {}
  T* allocate(size_t n) { return static_cast<T*>(MmAllocate(n * sizeof(T))); }
  void deallocate(T* p, size_t) noexcept { MmDeallocate(p); }
  template <class U> bool operator==(const MmAlloc<U>&) const noexcept { return true; }
  template <class U> bool operator!=(const MmAlloc<U>&) const noexcept { return false; }
};

template <class T> using vec = std::vector<T, MmAlloc<T>>;

}  // namespace jxlhip
