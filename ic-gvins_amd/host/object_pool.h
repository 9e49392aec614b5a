// Fixed-size block pools for the front-end's object graph (features, map points and their shared_ptr control blocks).
//
// The reference allocates every Feature / MapPoint with make_shared on the general heap (tracking/feature.h:52, mappoint.cc:50).  With
// hundreds of camera streams per host that is ~600 malloc/free pairs per frame, and — worse than their cost — blocks that come back cold
// and scattered.  PoolAllocator hands out blocks of one size class from per-thread LIFO free lists (the block freed last is the block
// reused first: still in cache), refilled from 64 KB slabs; a thread whose list grows past a bound (objects freed on another thread than
// the one that allocates) returns half of it to a shared list.  Memory is returned to the process, never to the OS (the graph's steady
// state is its peak).  Works with std::allocate_shared: the control block and the object share one pooled block.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

namespace icg {

template <size_t BlockBytes, size_t Align = 16> class BlockPool {
public:
    static void *allocate() {
        if (!tls_alive()) { // (see deallocate) no per-thread list any more: straight from the heap, block-sized and aligned
            void *q = nullptr;
            if (posix_memalign(&q, Align < sizeof(void *) ? sizeof(void *) : Align, kBlock) != 0) throw std::bad_alloc();
            return q;
        }
        Local &L = local();
        if (!L.head) refill(L);
        Node *n = L.head;
        L.head  = n->next;
        L.count--;
        return n;
    }
    static void deallocate(void *p) {
        if (!tls_alive()) { // this thread's free list is already destroyed (static-lifetime objects freed during thread / process exit)
            std::lock_guard<std::mutex> lock(shared().m);
            Node *n       = static_cast<Node *>(p);
            n->next       = shared().head;
            shared().head = n;
            return;
        }
        Local &L = local();
        Node *n  = static_cast<Node *>(p);
        n->next  = L.head;
        L.head   = n;
        if (++L.count > kLocalMax) spill(L);
    }

private:
    struct Node {
        Node *next;
    };
    struct Local {
        Node *head{nullptr};
        size_t count{0};
        Local() { tls_alive() = true; }
        ~Local() { // thread exit: hand the blocks back to the shared list; later frees on this thread go there directly
            tls_alive() = false;
            if (!head) return;
            std::lock_guard<std::mutex> lock(shared().m);
            while (head) {
                Node *n = head;
                head    = n->next;
                n->next = shared().head;
                shared().head = n;
            }
        }
    };
    struct Shared {
        std::mutex m;
        Node *head{nullptr};
    };
    static_assert(Align >= 16 && (Align & (Align - 1)) == 0 && Align <= 4096, "block alignment must be a power of two >= 16");
    static constexpr size_t kBlock    = (BlockBytes + Align - 1) / Align * Align;
    static constexpr size_t kSlab     = 64 * 1024;
    static constexpr size_t kLocalMax = 8192;
    // trivially destructible flag next to the list: true while this thread's Local exists (or has not been created yet: local()
    // creates it on first use), false once its destructor ran
    static bool &tls_alive() {
        static thread_local bool alive = true;
        return alive;
    }
    static Local &local() {
        static thread_local Local L;
        return L;
    }
    static Shared &shared() {
        static Shared *S = new Shared(); // intentionally leaked: threads may exit after static destruction started
        return *S;
    }
    static void refill(Local &L) {
        {
            std::lock_guard<std::mutex> lock(shared().m);
            size_t taken = 0;
            while (shared().head && taken < kLocalMax / 4) {
                Node *n       = shared().head;
                shared().head = n->next;
                n->next       = L.head;
                L.head        = n;
                taken++;
            }
            L.count += taken;
            if (taken) return;
        }
        void *raw = nullptr; // slabs are aligned to the block alignment (malloc only guarantees 16: an AVX Eigen member needs 32)
        if (posix_memalign(&raw, Align, kSlab) != 0 || !raw) throw std::bad_alloc();
        char *slab = static_cast<char *>(raw);
        const size_t n = kSlab / kBlock;
        for (size_t k = n; k-- > 0;) { // ascending addresses come off the list first: a burst of allocations walks memory forwards
            Node *b = reinterpret_cast<Node *>(slab + k * kBlock);
            b->next = L.head;
            L.head  = b;
        }
        L.count += n;
    }
    static void spill(Local &L) {
        std::lock_guard<std::mutex> lock(shared().m);
        for (size_t k = 0; k < kLocalMax / 2 && L.head; k++) {
            Node *n       = L.head;
            L.head        = n->next;
            n->next       = shared().head;
            shared().head = n;
            L.count--;
        }
    }
};

// std::allocator-compatible front for single-object allocations (allocate_shared, node-based containers); anything else goes to the heap
template <typename T> struct PoolAllocator {
    typedef T value_type;
    // blocks carry the alignment of the type (ICG_REFERENCE_TYPES: fixed-size vectorizable Eigen members may ask for 32 bytes)
    typedef BlockPool<sizeof(T), (alignof(T) > 16 ? alignof(T) : 16)> Pool;
    PoolAllocator() = default;
    template <typename U> PoolAllocator(const PoolAllocator<U> &) {}
    T *allocate(size_t n) {
        if (n == 1) return static_cast<T *>(Pool::allocate());
        return static_cast<T *>(::operator new(n * sizeof(T)));
    }
    void deallocate(T *p, size_t n) {
        if (n == 1)
            Pool::deallocate(p);
        else
            ::operator delete(p);
    }
    template <typename U> bool operator==(const PoolAllocator<U> &) const { return true; }
    template <typename U> bool operator!=(const PoolAllocator<U> &) const { return false; }
};

} // namespace icg
