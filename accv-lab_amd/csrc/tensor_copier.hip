// H3 — multi-tensor copier, native side (gfx950 / ROCm host runtime).
//
//   * accv_mtc_plan:     packs small host tensors into <= max_chunk aligned byte chunks.  Integer-exact restatement
//                        target: packages/multi_tensor_copier/accvlab/multi_tensor_copier/csrc/multi_tensor_copier.cpp
//                        :419-433 (bucket key), :481-507 (candidate rule is applied by the caller), :513-549 (offset
//                        layout), :553-590 (enable only if >= 2 packed).
//   * pinned arena:      size-class cache over hipHostMalloc — the reference allocates pinned memory per call
//                        (:597-641); hipHostMalloc is far too slow for that, so buffers are recycled.
//   * accv_mtc_stage_h2d: parallel memcpy of the leaves into the pinned chunk (the reference's
//                        fill_cpu_staging_buffers, :647-679) pipelined per chunk with ONE hipMemcpyAsync per chunk
//                        (enqueue_packed_transfer, :683-730) on the caller-supplied side stream.
//   * accv_mtc_coalesce: a single device kernel that gathers many small device tensors into one contiguous
//                        device buffer (no reference counterpart: the reference copies D2H/D2D per tensor,
//                        :775-820) so that ONE hipMemcpyAsync moves them.
// No torch types; the python host (accvlab/multi_tensor_copier) owns tensors, streams and events.
#include <hip/hip_runtime.h>
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "accv_common.h"

namespace {

inline int64_t round_up(int64_t x, int64_t a)
{
    if (a <= 1) return x;
    const int64_t rem = x % a;
    return rem == 0 ? x : x + (a - rem);
}

inline int bucket_of(int64_t required_align)  // 16, 8, 4, 2, 1 -> 0..4 (rounded DOWN to the bucket)
{
    if (required_align >= 16) return 0;
    if (required_align >= 8) return 1;
    if (required_align >= 4) return 2;
    if (required_align >= 2) return 3;
    return 4;
}

// Process-wide singletons live on the heap and are never destroyed: a function-local static's destructor would join its
// threads at process exit, after the HIP runtime may already be gone, and a forked child must be able to start
// over — the parent's threads do not exist there and a mutex another thread held at fork() time stays locked for ever, so
// the pthread_atfork child handler below simply forgets the parent's instances (leaked; fresh ones are created on demand).
// accv_mtc_shutdown() is the orderly end: python calls it from atexit, i.e. before HIP teardown.
template <class T>
struct Leaked {
    std::atomic<T*> ptr{nullptr};
    T& get()
    {
        T* p = ptr.load(std::memory_order_acquire);
        if (!p) {
            T* fresh = new T;
            if (ptr.compare_exchange_strong(p, fresh, std::memory_order_acq_rel))
                p = fresh;
            else
                delete fresh;   // lost the race: another thread installed its instance first
        }
        return *p;
    }
    void forget() { ptr.store(nullptr, std::memory_order_release); }
};
void register_fork_handler();

// ------------------------------------------------------------------------------------------ worker pool
class WorkerPool {
public:
    static Leaked<WorkerPool>& slot()
    {
        static Leaked<WorkerPool> s;
        return s;
    }
    static WorkerPool& instance()
    {
        register_fork_handler();
        return slot().get();
    }
    int size() const { return (int)workers_.size(); }
    struct Job {
        const std::function<void(int)>* fn;
        int tasks;
        std::atomic<int> next{0};
        std::atomic<int> finished{0};
        std::mutex done_mutex;
        std::condition_variable done_cv;
    };

    // runs fn(t) for t in [0, tasks) on the pool (the calling thread helps) and waits for completion
    void parallel(int tasks, const std::function<void(int)>& fn)
    {
        if (tasks <= 0) return;
        if (tasks == 1 || workers_.empty()) {
            for (int t = 0; t < tasks; ++t) fn(t);
            return;
        }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->tasks = tasks;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            queue_.push_back(job);
        }
        cv_.notify_all();
        run(*job);
        std::unique_lock<std::mutex> lock(job->done_mutex);
        job->done_cv.wait(lock, [&] { return job->finished.load() >= job->tasks; });
        std::lock_guard<std::mutex> qlock(mutex_);
        queue_.erase(std::remove(queue_.begin(), queue_.end(), job), queue_.end());
    }

    WorkerPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        int n = (int)std::min<unsigned>(hw ? hw : 1, 16);
        for (int i = 0; i + 1 < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }

private:
    static void run(Job& job)
    {
        for (;;) {
            const int t = job.next.fetch_add(1);
            if (t >= job.tasks) break;
            (*job.fn)(t);
            if (job.finished.fetch_add(1) + 1 >= job.tasks) {
                std::lock_guard<std::mutex> lock(job.done_mutex);
                job.done_cv.notify_all();
            }
        }
    }
    void loop()
    {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                cv_.wait(lock, [&] {
                    if (stop_) return true;
                    for (auto& j : queue_)
                        if (j->next.load() < j->tasks) return true;
                    return false;
                });
                if (stop_) return;
                for (auto& j : queue_)
                    if (j->next.load() < j->tasks) {
                        job = j;
                        break;
                    }
            }
            if (job) run(*job);
        }
    }

    std::vector<std::thread> workers_;
    std::vector<std::shared_ptr<Job>> queue_;
    std::mutex mutex_;
    std::condition_variable cv_;
    bool stop_ = false;
};

// ------------------------------------------------------------------------------------------ pinned arena
class PinnedArena {
public:
    static Leaked<PinnedArena>& slot()
    {
        static Leaked<PinnedArena> s;
        return s;
    }
    static PinnedArena& instance()
    {
        register_fork_handler();
        return slot().get();
    }
    void* acquire(size_t bytes)
    {
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> lock(mutex_);
            auto it = free_.find(cls);
            if (it != free_.end() && !it->second.empty()) {
                void* p = it->second.back();
                it->second.pop_back();
                live_[p] = cls;
                return p;
            }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, cls, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        std::lock_guard<std::mutex> lock(mutex_);
        live_[p] = cls;
        total_ += cls;
        return p;
    }
    void release(void* p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lock(mutex_);
        auto it = live_.find(p);
        if (it == live_.end()) return;
        free_[it->second].push_back(p);
        live_.erase(it);
    }
    void trim()
    {
        std::lock_guard<std::mutex> lock(mutex_);
        for (auto& kv : free_)
            for (void* p : kv.second) {
                (void)hipHostFree(p);
                total_ -= kv.first;
            }
        free_.clear();
    }
    size_t total() const { return total_; }

private:
    static size_t size_class(size_t bytes)
    {
        size_t c = 64 * 1024;
        while (c < bytes) c <<= 1;  // powers of two from 64 KiB: at most 2x slack, few distinct classes
        return c;
    }
    std::mutex mutex_;
    std::map<size_t, std::vector<void*>> free_;
    std::map<void*, size_t> live_;
    size_t total_ = 0;
};

// ------------------------------------------------------------------------------------------ device coalescing kernel
struct CopyItem {
    const void* src;
    long long dst_offset;
    long long nbytes;
};

// one workgroup per item (grid-stride over items); 16-byte vectors when both sides are 16-byte aligned
__global__ __launch_bounds__(256) void coalesce_kernel(const CopyItem* __restrict__ items, long long n_items,
                                                       unsigned char* __restrict__ packed, int scatter)
{
    for (long long it = blockIdx.x; it < n_items; it += gridDim.x) {
        const CopyItem d = items[it];
        const unsigned char* s = scatter ? packed + d.dst_offset : static_cast<const unsigned char*>(d.src);
        unsigned char* o = scatter ? const_cast<unsigned char*>(static_cast<const unsigned char*>(d.src)) : packed + d.dst_offset;
        const bool vec = (((uintptr_t)s | (uintptr_t)o) & 15u) == 0;
        long long done = 0;
        if (vec) {
            const long long n16 = d.nbytes >> 4;
            const uint4* s4 = reinterpret_cast<const uint4*>(s);
            uint4* o4 = reinterpret_cast<uint4*>(o);
            for (long long i = threadIdx.x; i < n16; i += blockDim.x) o4[i] = s4[i];
            done = n16 << 4;
        }
        for (long long i = done + threadIdx.x; i < d.nbytes; i += blockDim.x) o[i] = s[i];
    }
}

// ------------------------------------------------------------------------------------------ native orchestration
extern "C" int accv_mtc_stage_h2d(long long, const void* const*, const long long*, const long long*, const long long*,
                                  long long, const long long*, void* const*, void* const*, const long long*, void*, int);

struct AsyncStage {
    std::vector<const void*> src;
    std::vector<long long> nbytes, offset, order, item_begin, chunk_bytes;
    std::vector<void*> staging, device;
    long long n_items = 0, n_chunks = 0;
    void* stream = nullptr;
    int threads = 0, device_index = 0;
    // result
    bool done = false;
    int status = 0;
    char error[512] = {0};
    void release_arguments()
    {
        for (auto* v : {&nbytes, &offset, &order, &item_begin, &chunk_bytes}) std::vector<long long>().swap(*v);
        std::vector<const void*>().swap(src);
        std::vector<void*>().swap(staging);
        std::vector<void*>().swap(device);
    }
};

// ONE persistent thread executes the queued jobs in order (a job already fans its memcpy out over the worker pool).
// A ticket stays valid until it was waited for once; tickets that are never waited for (a handle that failed before its
// wait, or was only polled) do not pile up: a finished job drops its argument vectors at once, and submit() forgets finished
// tickets that are more than kKeepTickets submissions old.
class Orchestrator {
public:
    static constexpr long long kKeepTickets = 1024;
    static Leaked<Orchestrator>& slot()
    {
        static Leaked<Orchestrator> s;
        return s;
    }
    static Orchestrator& instance()
    {
        register_fork_handler();
        return slot().get();
    }
    long long submit(const std::shared_ptr<AsyncStage>& job)
    {
        std::lock_guard<std::mutex> lock(mutex_);
        if (stop_) return -1;   // after accv_mtc_shutdown
        if (!started_) {
            started_ = true;
            thread_ = std::thread([this] { loop(); });
        }
        const long long id = ++next_id_;
        for (auto it = jobs_.begin(); it != jobs_.end() && it->first <= id - kKeepTickets;)
            it = it->second->done ? jobs_.erase(it) : std::next(it);
        jobs_[id] = job;
        queue_.push_back(job);
        cv_.notify_all();
        return id;
    }
    // block = true: wait for the job, forget the ticket, return its status; block = false: 1 done / 0 running
    int wait(long long ticket, bool block)
    {
        std::unique_lock<std::mutex> lock(mutex_);
        auto it = jobs_.find(ticket);
        if (it == jobs_.end()) return accv::fail(ACCV_EINVAL, "mtc async: unknown ticket %lld", ticket);
        std::shared_ptr<AsyncStage> job = it->second;
        if (!block) return job->done ? 1 : 0;
        done_cv_.wait(lock, [&] { return job->done; });
        jobs_.erase(ticket);   // (the iterator may be stale: submit() can have swept older tickets meanwhile)
        if (job->status != ACCV_OK) snprintf(accv::error_buffer(), 512, "%s", job->error);
        return job->status;
    }
    size_t tickets_held()
    {
        std::lock_guard<std::mutex> lock(mutex_);
        return jobs_.size();
    }
    // drain the queue, stop and join the thread (idempotent); later submissions are refused
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        cv_.notify_all();
        if (thread_.joinable()) thread_.join();
    }
    Orchestrator() = default;
    ~Orchestrator() { shutdown(); }

private:
    void loop()
    {
        int current_device = -1;
        for (;;) {
            std::shared_ptr<AsyncStage> job;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                cv_.wait(lock, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty()) return;   // stop requested and nothing left to run (queued jobs are drained first)
                job = queue_.front();
                queue_.erase(queue_.begin());
            }
            int rc = ACCV_OK;
            if (job->device_index >= 0 && job->device_index != current_device) {   // < 0: staging only, no device involved
                if (hipSetDevice(job->device_index) != hipSuccess) {
                    (void)hipGetLastError();
                    rc = accv::fail(ACCV_ELAUNCH, "mtc async: hipSetDevice(%d) failed", job->device_index);
                } else {
                    current_device = job->device_index;
                }
            }
            if (rc == ACCV_OK)
                rc = accv_mtc_stage_h2d(job->n_items, job->src.data(), job->nbytes.data(), job->offset.data(), job->order.data(),
                                        job->n_chunks, job->item_begin.data(), job->staging.data(), job->device.data(),
                                        job->chunk_bytes.data(), job->stream, job->threads);
            {
                std::lock_guard<std::mutex> lock(mutex_);
                job->status = rc;
                if (rc != ACCV_OK) snprintf(job->error, sizeof job->error, "%s", accv::error_buffer());   // this thread's message
                job->release_arguments();   // a ticket nobody waits for keeps ~600 bytes, not the copied argument vectors
                job->done = true;
            }
            done_cv_.notify_all();
        }
    }
    std::mutex mutex_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::shared_ptr<AsyncStage>> queue_;
    std::map<long long, std::shared_ptr<AsyncStage>> jobs_;
    std::thread thread_;
    long long next_id_ = 0;
    bool started_ = false, stop_ = false;
};

void forget_singletons_in_child()
{
    WorkerPool::slot().forget();
    PinnedArena::slot().forget();   // the parent's pinned blocks are not this process's to hand out
    Orchestrator::slot().forget();
}
void register_fork_handler()
{
    static std::atomic<bool> done{false};
    if (!done.exchange(true)) pthread_atfork(nullptr, nullptr, forget_singletons_in_child);
}

}  // namespace

extern "C" {

int accv_mtc_plan(long long n, const long long* nbytes, const int* elem_size, const unsigned char* candidate,
                  long long min_align, long long max_chunk_bytes, long long* out_offset, long long* out_chunk,
                  long long* out_chunk_sizes, long long* out_num_chunks)
{
    if (n < 0 || !out_num_chunks) return accv::fail(ACCV_EINVAL, "mtc_plan: bad arguments");
    *out_num_chunks = 0;
    if (n == 0) return ACCV_OK;
    if (!nbytes || !elem_size || !candidate || !out_offset || !out_chunk || !out_chunk_sizes)
        return accv::fail(ACCV_EINVAL, "mtc_plan: null array");
    min_align = std::max<long long>(1, min_align);
    std::vector<long long> order[5];
    std::vector<long long> req(n, 1);
    for (long long i = 0; i < n; ++i) {
        out_offset[i] = -1;
        out_chunk[i] = -1;
        if (!candidate[i]) continue;
        const long long es = std::max(1, elem_size[i]);
        req[i] = round_up(std::max<long long>(min_align, es), es);
        order[bucket_of(req[i])].push_back(i);
    }
    long long cursor = 0, chunk = 0, packed = 0, n_chunks = 0;
    for (int b = 0; b < 5; ++b) {
        for (long long i : order[b]) {
            long long at = round_up(cursor, req[i]);
            if (at + nbytes[i] > max_chunk_bytes && cursor > 0) {
                out_chunk_sizes[n_chunks++] = cursor;
                cursor = 0;
                ++chunk;
                at = 0;
            }
            out_offset[i] = at;
            out_chunk[i] = chunk;
            cursor = at + nbytes[i];
            ++packed;
        }
    }
    if (cursor > 0) out_chunk_sizes[n_chunks++] = cursor;
    if (packed < 2 || n_chunks == 0) {  // packing a single tensor buys nothing
        for (long long i = 0; i < n; ++i) {
            out_offset[i] = -1;
            out_chunk[i] = -1;
        }
        n_chunks = 0;
    }
    *out_num_chunks = n_chunks;
    return ACCV_OK;
}

void* accv_pinned_acquire(size_t bytes)
{
    void* p = PinnedArena::instance().acquire(bytes ? bytes : 1);
    if (!p) accv::fail(ACCV_ERUNTIME, "pinned arena: hipHostMalloc(%zu) failed", bytes);
    return p;
}

void accv_pinned_release(void* p) { PinnedArena::instance().release(p); }

void accv_pinned_trim(void) { PinnedArena::instance().trim(); }

size_t accv_pinned_total_bytes(void) { return PinnedArena::instance().total(); }

int accv_mtc_worker_count(void) { return WorkerPool::instance().size() + 1; }

/* Host staging + transfer.  For every chunk c (in order): memcpy the items with chunk_of[i] == c into
 * staging[c] + offset[i] using up to `threads` workers, then hipMemcpyAsync(device[c], staging[c], chunk_bytes[c])
 * host->device on `stream`.  Items must be grouped so that item_begin[c]..item_begin[c+1] index `order`.
 * device[c] == NULL skips the transfer (staging only).  Does not synchronise the stream. */
int accv_mtc_pack_host(long long n_items, const void* const* src, const long long* nbytes, const long long* offset,
                       void* dst, long long dst_bytes)
{
    if (n_items < 0) return accv::fail(ACCV_EINVAL, "mtc_pack_host: negative count");
    if (n_items == 0) return ACCV_OK;
    if (!src || !nbytes || !offset || !dst) return accv::fail(ACCV_EINVAL, "mtc_pack_host: null array");
    char* base = static_cast<char*>(dst);
    // plain loop on the calling thread, no HIP call and no pool: this runs inside forked DataLoader workers, where
    // neither the parent's worker threads nor its HIP context exist
    for (long long i = 0; i < n_items; ++i) {
        if (nbytes[i] < 0 || offset[i] < 0 || offset[i] + nbytes[i] > dst_bytes)
            return accv::fail(ACCV_EINVAL, "mtc_pack_host: item %lld does not fit the buffer", i);
        if (nbytes[i]) std::memcpy(base + offset[i], src[i], (size_t)nbytes[i]);
    }
    return ACCV_OK;
}

int accv_mtc_stage_h2d(long long n_items, const void* const* src, const long long* nbytes, const long long* offset,
                       const long long* order, long long n_chunks, const long long* item_begin, void* const* staging,
                       void* const* device, const long long* chunk_bytes, void* stream_, int threads)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_items < 0 || n_chunks < 0) return accv::fail(ACCV_EINVAL, "mtc_stage_h2d: negative count");
    if (n_chunks == 0) return ACCV_OK;
    if (!src || !nbytes || !offset || !order || !item_begin || !staging || !device || !chunk_bytes)
        return accv::fail(ACCV_EINVAL, "mtc_stage_h2d: null array");
    WorkerPool& pool = WorkerPool::instance();
    threads = std::max(1, std::min(threads > 0 ? threads : pool.size() + 1, pool.size() + 1));
    for (long long c = 0; c < n_chunks; ++c) {
        const long long lo = item_begin[c], hi = item_begin[c + 1];
        if (lo < 0 || hi < lo || hi > n_items) return accv::fail(ACCV_EINVAL, "mtc_stage_h2d: bad item range");
        char* base = static_cast<char*>(staging[c]);
        if (!base && hi > lo) return accv::fail(ACCV_EINVAL, "mtc_stage_h2d: null staging buffer");
        long long bytes = 0;
        for (long long k = lo; k < hi; ++k) bytes += nbytes[order[k]];
        // split the item range into `tasks` pieces of roughly equal bytes
        int tasks = (int)std::min<long long>(threads, std::max<long long>(1, bytes / (256 * 1024)));
        tasks = (int)std::min<long long>(tasks, std::max<long long>(1, hi - lo));
        std::vector<long long> cut(tasks + 1, hi);
        cut[0] = lo;
        if (tasks > 1) {
            long long acc = 0, next = 1;
            for (long long k = lo; k < hi && next < tasks; ++k) {
                acc += nbytes[order[k]];
                if (acc >= bytes * next / tasks) cut[next++] = k + 1;
            }
        }
        std::function<void(int)> work = [&](int t) {
            for (long long k = cut[t]; k < cut[t + 1]; ++k) {
                const long long i = order[k];
                std::memcpy(base + offset[i], src[i], (size_t)nbytes[i]);
            }
        };
        pool.parallel(tasks, work);
        if (device[c] && chunk_bytes[c] > 0) {
            hipError_t e = hipMemcpyAsync(device[c], base, (size_t)chunk_bytes[c], hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return accv::fail(ACCV_ELAUNCH, "mtc_stage_h2d: hipMemcpyAsync: %s", hipGetErrorString(e));
        }
    }
    return ACCV_OK;
}

/* ---- the same staging + transfer on a NATIVE orchestration thread (the reference's CopyThreadPool worker,
 * multi_tensor_copier.cpp:288-349, 863-883): the call copies its argument arrays, queues the job and returns a ticket at
 * once; a library thread (no Python, no GIL) runs accv_mtc_stage_h2d on `device_index`; accv_mtc_async_wait blocks until
 * the transfers of that job have been ENQUEUED on the stream (not completed: the caller records / synchronises a stream
 * event afterwards) and returns the job's status (its error text becomes the waiting thread's accv_last_error). */
int accv_mtc_stage_h2d_async(long long n_items, const void* const* src, const long long* nbytes, const long long* offset,
                             const long long* order, long long n_chunks, const long long* item_begin,
                             void* const* staging, void* const* device, const long long* chunk_bytes, void* stream,
                             int threads, int device_index, long long* ticket_out)
{
    if (!ticket_out) return accv::fail(ACCV_EINVAL, "mtc_stage_h2d_async: null ticket pointer");
    if (n_items < 0 || n_chunks < 0) return accv::fail(ACCV_EINVAL, "mtc_stage_h2d_async: negative count");
    if (n_chunks > 0 && (!src || !nbytes || !offset || !order || !item_begin || !staging || !device || !chunk_bytes))
        return accv::fail(ACCV_EINVAL, "mtc_stage_h2d_async: null array");
    auto job = std::make_shared<AsyncStage>();
    job->src.assign(src, src + (n_chunks ? n_items : 0));
    job->nbytes.assign(nbytes, nbytes + (n_chunks ? n_items : 0));
    job->offset.assign(offset, offset + (n_chunks ? n_items : 0));
    job->order.assign(order, order + (n_chunks ? n_items : 0));
    job->item_begin.assign(item_begin, item_begin + (n_chunks ? n_chunks + 1 : 0));
    job->staging.assign(staging, staging + n_chunks);
    job->device.assign(device, device + n_chunks);
    job->chunk_bytes.assign(chunk_bytes, chunk_bytes + n_chunks);
    job->n_items = n_items;
    job->n_chunks = n_chunks;
    job->stream = stream;
    job->threads = threads;
    job->device_index = device_index;
    const long long ticket = Orchestrator::instance().submit(job);
    if (ticket < 0) return accv::fail(ACCV_ERUNTIME, "mtc_stage_h2d_async: the orchestration thread was shut down");
    *ticket_out = ticket;
    return ACCV_OK;
}

int accv_mtc_async_wait(long long ticket) { return Orchestrator::instance().wait(ticket, true); }

/* Orderly end of the native orchestration thread: runs what is still queued, then stops and joins the thread; later
 * accv_mtc_stage_h2d_async calls fail.  The python package registers it with atexit so that no job is inside
 * hipMemcpyAsync while the HIP runtime is torn down (the reference's CopyThreadPool joins its workers in its destructor,
 * multi_tensor_copier.cpp:300-312).  Idempotent. */
void accv_mtc_shutdown(void)
{
    if (Orchestrator* o = Orchestrator::slot().ptr.load(std::memory_order_acquire)) o->shutdown();
}

/* Tickets the orchestrator still remembers (diagnostics / tests: bounded even when handles are never waited for). */
long long accv_mtc_async_tickets_held(void) { return (long long)Orchestrator::instance().tickets_held(); }

/* 1 = the job has finished (successfully or not; its status is what accv_mtc_async_wait returns), 0 = still running,
 * negative = unknown ticket. */
int accv_mtc_async_poll(long long ticket) { return Orchestrator::instance().wait(ticket, false); }

/* Device-side coalescing: items[k] = {device src pointer, byte offset inside `packed`, nbytes}; `items` must be
 * readable from the device (device memory or pinned host memory).  scatter == 0 gathers src -> packed + offset,
 * scatter != 0 copies packed + offset -> src (used to fan a packed buffer out again). */
int accv_mtc_coalesce(const void* items, long long n_items, void* packed, int scatter, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_items < 0) return accv::fail(ACCV_EINVAL, "mtc_coalesce: negative count");
    if (n_items == 0) return ACCV_OK;
    if (!items || !packed) return accv::fail(ACCV_EINVAL, "mtc_coalesce: null pointer");
    const unsigned grid = (unsigned)std::min<long long>(n_items, 256 * 16);
    hipLaunchKernelGGL(coalesce_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const CopyItem*>(items), n_items,
                       static_cast<unsigned char*>(packed), scatter);
    return accv::check_launch("mtc_coalesce");
}

/* Thin wrappers so the python host never needs another HIP binding. */
int accv_memcpy_async(void* dst, const void* src, size_t bytes, int kind, void* stream)
{
    hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost
                    : kind == 3 ? hipMemcpyDeviceToDevice : hipMemcpyDefault;
    if (bytes == 0) return ACCV_OK;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, k, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return accv::fail(ACCV_ELAUNCH, "memcpy_async: %s", hipGetErrorString(e));
    return ACCV_OK;
}
}
