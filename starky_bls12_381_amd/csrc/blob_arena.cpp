// Page-locked, recycled proof blobs (blob_arena.h).
#include "blob_arena.h"

#include <hip/hip_runtime_api.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

namespace starkhip {
namespace {

struct Slot {
    void* p;
    size_t cap;
    const void* owner;
    bool busy, orphan;
};
std::mutex g_mu;
std::vector<Slot> g_slots;  // a few dozen at most: linear searches
unsigned long g_taken = 0, g_missed = 0;

}  // namespace

uint64_t* blob_alloc(size_t bytes) {
    {
        std::lock_guard<std::mutex> g(g_mu);
        Slot* best = nullptr;
        for (Slot& s : g_slots)
            if (!s.busy && !s.orphan && s.cap >= bytes && (!best || s.cap < best->cap)) best = &s;
        if (best) {
            best->busy = true;
            g_taken++;
            return (uint64_t*)best->p;
        }
        g_missed++;
    }
    return (uint64_t*)malloc(bytes ? bytes : 1);
}

void blob_free(void* p) {
    if (!p) return;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_slots.size(); i++)
            if (g_slots[i].p == p) {
                if (!g_slots[i].orphan) {
                    g_slots[i].busy = false;
                    return;
                }
                g_slots.erase(g_slots.begin() + i);
                lk.unlock();
                (void)hipHostFree(p);
                return;
            }
    }
    free(p);
}

int blob_arena_add(const void* owner, size_t bytes, unsigned count) {
    for (unsigned i = 0; i < count; i++) {
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
        if (e != hipSuccess) return (int)e;
        std::lock_guard<std::mutex> g(g_mu);
        g_slots.push_back(Slot{p, bytes, owner, false, false});
    }
    return 0;
}

void blob_arena_drop(const void* owner) {
    std::vector<void*> gone;
    {
        std::lock_guard<std::mutex> g(g_mu);
        for (size_t i = 0; i < g_slots.size();) {
            Slot& s = g_slots[i];
            if (s.owner != owner) {
                i++;
            } else if (s.busy) {
                s.orphan = true;
                s.owner = nullptr;
                i++;
            } else {
                gone.push_back(s.p);
                g_slots.erase(g_slots.begin() + i);
            }
        }
    }
    for (void* p : gone) (void)hipHostFree(p);
}

BlobArenaStats blob_arena_stats() {
    std::lock_guard<std::mutex> g(g_mu);
    BlobArenaStats st = {g_slots.size(), 0, 0, g_taken, g_missed};
    for (const Slot& s : g_slots) {
        st.busy += s.busy;
        st.bytes += s.cap;
    }
    return st;
}

}  // namespace starkhip
