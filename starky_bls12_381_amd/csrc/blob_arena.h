// Page-locked, recycled proof blobs.
//
// A proof is 21 .. 69 MB (84 query rounds x C leaf words), written by ONE device-to-host copy at the end of prove().  From
// malloc that block is a fresh mmap every time (glibc caps M_MMAP_THRESHOLD at 32 MB), i.e. first-touch page faults on every
// page and a pageable copy staged through the runtime's bounce buffers: 5 .. 6 ms of a proof's "queries" phase (FinalExp 6.1 ms,
// MillerLoop 5.8 ms, profiles/r03_s_*) against ~ 1.5 ms for the same copy into page-locked memory.  hipHostMalloc, however, waits
// for every stream of the device, so it must not happen while proofs are in flight: a context RESERVES blobs when a pool warms
// it up (ctx_reserve), prove() takes one if one is idle and falls back to malloc otherwise, and starkhip_free() hands it back.
// The arena is process-wide (a blob outlives the prove() call and may be freed from any thread); blobs belong to the context that
// reserved them and are released to the system when that context is destroyed (at once if idle, otherwise when they come back).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace starkhip {

uint64_t* blob_alloc(size_t bytes);  // an idle arena blob of at least `bytes` (the smallest that fits), else malloc; nullptr: out of memory
void blob_free(void* p);             // an arena blob goes back to the arena, anything else to free(); nullptr is fine
// `count` new page-locked blobs of `bytes` owned by `owner`.  Device-wide synchronisation inside: warm-up only.  0 or a hipError_t.
int blob_arena_add(const void* owner, size_t bytes, unsigned count);
void blob_arena_drop(const void* owner);  // the owner goes away: its idle blobs are released now, its busy ones when they are freed
struct BlobArenaStats {
    size_t blobs, busy, bytes;
    unsigned long taken, missed;  // blob_alloc calls served from the arena / by malloc
};
BlobArenaStats blob_arena_stats();

}  // namespace starkhip
