// pnx_host_pipeline.hpp -- the host-side orchestration of PNX_MEM_HOST calls (threads, flags, hand-over of chunks and
// granules between them), free of HIP types: pnx_api.hip instantiates it on the HIP runtime, tests/host_stub/ builds the same
// code for the CPU against a stub device and runs it under ThreadSanitizer and AddressSanitizer (tests/test_host_sanitizers.py).
//
//   run_pipeline_t<B>   the chunk ring: IN / LAUNCH / OUT / page-touch stages over n_slots device slots
//   run_streamed        the state machine around ONE persistent kernel that consumes a volume while it is being uploaded and
//                       whose results are downloaded granule by granule while it still runs (curve fit from host arrays)
//
// Hot path served: the reference's fitters hand their solvers whole numpy arrays (fitters/pixelwise.py:91-96); SURVEY 8(d)
// defines the metric as the C-ABI call including the transfers.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pnx.h"

namespace pnx {
// records a printf-style message for pnx_last_error() and returns `code` (pnx_api.hip; the stub defines its own)
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// the calling thread's last message (the failing helper thread's message travels to the caller through it)
const char *last_error_text();

// ---- host-staging pipeline ------------------------------------------------------------------------------------
// PNX_MEM_HOST calls hand over pageable numpy memory.  The volume is cut into chunks of voxels that flow through a
// ring of device slots, one stage per host thread:
//   IN      blocking H2D copy of chunk k into slot k % S (waits for the slot to be drained)
//   LAUNCH  (the calling thread) enqueues the kernels of chunk k, alternating between the kernel streams so that the
//           drain tail of one chunk overlaps the start of the next; records an event
//   OUT     waits for the event, blocking D2H copy of chunk k into the caller's arrays
//   TOUCH   helper threads take the first-touch page faults of the freshly allocated result arrays ahead of OUT
//           (42 ms per GB when they are taken serially inside the D2H copy)
// so H2D, compute, D2H and the page faults overlap; the call returns when every stage has drained.
template <class Stream> struct PipeOpsT {
    std::function<int(int k, int slot, Stream st)> h2d, launch, d2h;
    std::function<void(int k)> touch;  // may be empty
};

// B (the runtime): stream_t, event_t, and static
//   bool bind_device(int)                       the calling helper thread will talk to this device
//   bool stream_create(stream_t *, bool kernel) kernel streams get the lowest priority (a queue of their own, see run_streamed)
//   void stream_destroy(stream_t)   bool stream_sync(stream_t)
//   bool event_create(event_t *)    void event_destroy(event_t)   bool event_record(event_t, stream_t)   bool event_sync(event_t)
template <class B>
int run_pipeline_t(int n_chunks, int n_slots, int k_streams, int touchers, int device, typename B::stream_t user_stream,
                   const PipeOpsT<typename B::stream_t> &ops, bool trace) {
    typedef typename B::stream_t stream_t;
    typedef typename B::event_t event_t;
    if (n_chunks == 1) {  // small batch: everything on the caller's stream, no threads
        int rc = ops.h2d(0, 0, user_stream);
        if (!rc) rc = ops.launch(0, 0, user_stream);
        if (rc) return rc;
        if (ops.touch) ops.touch(0);
        if (!B::stream_sync(user_stream)) return set_error(PNX_ERR_HIP, "kernel of the call failed");
        rc = ops.d2h(0, 0, user_stream);
        if (rc) return rc;
        if (!B::stream_sync(user_stream)) return set_error(PNX_ERR_HIP, "D2H of the call failed");
        return PNX_OK;
    }
    if (user_stream && !B::stream_sync(user_stream)) return set_error(PNX_ERR_HIP, "the caller's stream reports an error");
    const auto t_call = std::chrono::steady_clock::now();
    auto now = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    std::vector<double> t_in(n_chunks), t_launch(n_chunks), t_kdone(n_chunks), t_out(n_chunks), t_touch(n_chunks);
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        int in_done = 0, launched = 0, drained = 0;
        int code = PNX_OK;
        std::string msg;
        bool failed = false;
    } sh;
    std::atomic<bool> stop(false);
    auto fail = [&](int code) {
        std::lock_guard<std::mutex> lk(sh.mu);
        if (!sh.failed) {
            sh.failed = true;
            sh.code = code;
            sh.msg = last_error_text();  // the failing thread's message
        }
        stop.store(true);
        sh.cv.notify_all();
    };
    stream_t s_in = stream_t(), s_out = stream_t(), s_k[4] = {stream_t(), stream_t(), stream_t(), stream_t()};
    std::vector<event_t> ev(n_chunks, event_t());
    auto cleanup = [&]() {
        if (s_in) B::stream_destroy(s_in);
        if (s_out) B::stream_destroy(s_out);
        for (auto &q : s_k)
            if (q) B::stream_destroy(q);
        for (auto &e : ev)
            if (e) B::event_destroy(e);
    };
    {
        bool ok = B::stream_create(&s_in, false) && B::stream_create(&s_out, false);
        for (int i = 0; i < k_streams && ok; ++i) ok = B::stream_create(&s_k[i], true);
        for (int k = 0; k < n_chunks && ok; ++k) ok = B::event_create(&ev[k]);
        if (!ok) {
            cleanup();
            return set_error(PNX_ERR_HIP, "pipeline stream/event setup failed");
        }
    }
    // claim[k]: 0 untouched, 1 being touched / touched by a helper
    std::vector<std::atomic<int>> claim(n_chunks), touched(n_chunks);
    for (int k = 0; k < n_chunks; ++k) {
        claim[k].store(0);
        touched[k].store(0);
    }
    std::vector<std::thread> th;
    th.emplace_back([&]() {  // IN
        if (!B::bind_device(device)) return fail(set_error(PNX_ERR_HIP, "hipSetDevice failed (IN thread)"));
        for (int k = 0; k < n_chunks; ++k) {
            {
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.cv.wait(lk, [&] { return sh.failed || sh.drained > k - n_slots; });
                if (sh.failed) return;
            }
            int rc = ops.h2d(k, k % n_slots, s_in);
            if (!rc && !B::stream_sync(s_in)) rc = set_error(PNX_ERR_HIP, "H2D of chunk %d failed", k);
            if (rc) return fail(rc);
            t_in[k] = now();
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.in_done = k + 1;
            sh.cv.notify_all();
        }
    });
    th.emplace_back([&]() {  // OUT
        if (!B::bind_device(device)) return fail(set_error(PNX_ERR_HIP, "hipSetDevice failed (OUT thread)"));
        for (int k = 0; k < n_chunks; ++k) {
            {
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.cv.wait(lk, [&] { return sh.failed || sh.launched > k; });
                if (sh.failed) return;
            }
            if (ops.touch) {
                int expect = 0;
                if (claim[k].compare_exchange_strong(expect, 1)) {  // no helper got here yet: touch it ourselves
                    ops.touch(k);
                    touched[k].store(1);
                } else {
                    while (!touched[k].load()) std::this_thread::yield();
                }
            }
            int rc = PNX_OK;
            t_touch[k] = now();
            if (!B::event_sync(ev[k])) rc = set_error(PNX_ERR_HIP, "kernel of chunk %d failed", k);
            t_kdone[k] = now();
            if (!rc) rc = ops.d2h(k, k % n_slots, s_out);
            if (!rc && !B::stream_sync(s_out)) rc = set_error(PNX_ERR_HIP, "D2H of chunk %d failed", k);
            if (rc) return fail(rc);
            t_out[k] = now();
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.drained = k + 1;
            sh.cv.notify_all();
        }
    });
    if (ops.touch)
        for (int t = 0; t < touchers; ++t)
            th.emplace_back([&]() {
                for (int k = 0; k < n_chunks; ++k) {
                    if (stop.load()) return;
                    int expect = 0;
                    if (claim[k].compare_exchange_strong(expect, 1)) {
                        ops.touch(k);
                        touched[k].store(1);
                    }
                }
            });
    // LAUNCH stage on the calling thread
    for (int k = 0; k < n_chunks; ++k) {
        {
            std::unique_lock<std::mutex> lk(sh.mu);
            sh.cv.wait(lk, [&] { return sh.failed || sh.in_done > k; });
            if (sh.failed) break;
        }
        stream_t st = s_k[k % k_streams];
        int rc = ops.launch(k, k % n_slots, st);
        if (!rc && !B::event_record(ev[k], st)) rc = set_error(PNX_ERR_HIP, "hipEventRecord failed");
        if (rc) {
            fail(rc);
            break;
        }
        t_launch[k] = now();
        std::lock_guard<std::mutex> lk(sh.mu);
        sh.launched = k + 1;
        sh.cv.notify_all();
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < k_streams; ++i) (void)B::stream_sync(s_k[i]);  // nothing of ours may outlive the call
    cleanup();
    if (trace)
        for (int k = 0; k < n_chunks; ++k)
            fprintf(stderr, "[pnx host] chunk %d: h2d_done %.1f launched %.1f out_ready %.1f kernel_done %.1f d2h_done %.1f ms\n", k, t_in[k],
                    t_launch[k], t_touch[k], t_kdone[k], t_out[k]);
    if (sh.failed) return set_error(sh.code, "%s", sh.msg.c_str());
    return PNX_OK;
}

// ---- one streamed kernel per call ------------------------------------------------------------------------------
// The device side of a streamed call as the orchestration sees it.  Every function may be called from the thread named;
// functions that return int return PNX_OK or an error code with the message set (set_error).
struct StreamedOps {
    std::function<bool()> bind_device;            // any helper thread, once
    std::function<int(int i)> upload_piece;       // IN: enqueue upload piece i and, behind it on the same stream, the watermark move
    std::function<int()> upload_sync;             // IN: wait for the upload stream
    std::function<bool()> first_piece_landed;     // caller: the first watermark move has executed on the device
    std::function<bool(int g)> granule_ready;     // OUT: completion flag of granule g, raised by the kernel (acquire)
    std::function<int(int g, int ot)> download;   // OUT thread ot: epilogue, copies and the wait for them, granule g
    std::function<void(int g)> touch;             // first-touch the result pages of granule g
    std::function<void()> abort_kernel;           // any thread: the kernel's lanes stop waiting for the watermark and leave
    std::function<int()> kernel_state;            // caller: 0 running, 1 ended, < 0 ended with an error (message set)
    std::function<int()> kernel_wait;             // caller: block until the kernel has ended; PNX_OK or an error code
};
struct StreamedTimes {  // filled when tracing
    std::vector<double> t_in, t_flag, t_out;
    double t_kernel = 0;
};

// Returns PNX_OK, an error code (message set), and *stalled = true when the watermark did not move within `stall_ms` after the
// launch (or a helper failed before the kernel could see data): the caller then runs the call through the chunk ring.
//
// Who stops whom.  The kernel never waits for another kernel, only for the watermark, and that wait ends on the abort word.
// ANY failure on the host side (upload, epilogue, download) raises the abort word at once, so the kernel drains in
// microseconds instead of spinning to its poll limit; a watermark that has not moved `stall_ms` after the launch (another
// library's streams sharing the hardware queue of the upload) does the same.
inline int run_streamed(int n_in, int n_gran, int n_out, int touchers, double stall_ms, int test_delay_ms, const StreamedOps &ops,
                        bool *stalled, StreamedTimes *times, std::function<double()> now) {
    *stalled = false;
    std::atomic<bool> failed(false), kernel_done(false), give_up(false);
    std::mutex err_mu;
    int err_code = PNX_OK;
    std::string err_msg;
    auto fail = [&](int code) {
        {
            std::lock_guard<std::mutex> lk(err_mu);
            if (!failed.load()) {
                err_code = code;
                err_msg = last_error_text();
                failed.store(true);
            }
        }
        ops.abort_kernel();  // whoever fails tells the kernel: nothing waits for a watermark that will not come
    };
    std::vector<std::atomic<int>> claim(n_gran), touched(n_gran);
    for (int g = 0; g < n_gran; ++g) {
        claim[g].store(0);
        touched[g].store(0);
    }
    if (times) {
        times->t_in.assign(n_in, 0.0);
        times->t_flag.assign(n_gran, 0.0);
        times->t_out.assign(n_gran, 0.0);
    }
    std::vector<std::thread> th;
    th.emplace_back([&]() {  // IN
        auto body = [&]() -> int {
            if (!ops.bind_device()) return set_error(PNX_ERR_HIP, "hipSetDevice failed (IN thread)");
            if (test_delay_ms) {  // tests: a stalled upload (sleeps in slices so that a call that gave up does not wait it out)
                const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(test_delay_ms);
                while (std::chrono::steady_clock::now() < until && !failed.load() && !give_up.load())
                    std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            for (int i = 0; i < n_in && !failed.load() && !give_up.load(); ++i) {
                const int r = ops.upload_piece(i);
                if (r) return r;
                if (times) times->t_in[i] = now();
            }
            return ops.upload_sync();
        };
        const int r = body();
        if (r) fail(r);
    });
    for (int ot = 0; ot < n_out; ++ot) th.emplace_back([&, ot]() {  // OUT
        auto body = [&]() -> int {
            if (!ops.bind_device()) return set_error(PNX_ERR_HIP, "hipSetDevice failed (OUT thread)");
            for (int g = ot; g < n_gran; g += n_out) {
                int expect = 0;
                if (claim[g].compare_exchange_strong(expect, 1)) {  // no helper got here yet: touch it ourselves
                    ops.touch(g);
                    touched[g].store(1);
                } else {
                    while (!touched[g].load()) std::this_thread::yield();
                }
                for (;;) {
                    if (ops.granule_ready(g)) break;
                    if (failed.load() || give_up.load()) return PNX_OK;
                    if (kernel_done.load()) {  // the kernel raises every flag before it ends
                        if (ops.granule_ready(g)) break;
                        if (failed.load() || give_up.load()) return PNX_OK;
                        return set_error(PNX_ERR_HIP, "streamed curve fit: kernel ended without completing granule %d", g);
                    }
                    std::this_thread::yield();
                }
                if (times) times->t_flag[g] = now();
                const int r = ops.download(g, ot);
                if (r) return r;
                if (times) times->t_out[g] = now();
            }
            return PNX_OK;
        };
        const int r = body();
        if (r) fail(r);
    });
    for (int t = 0; t < touchers; ++t)
        th.emplace_back([&]() {
            for (int g = 0; g < n_gran && !failed.load() && !give_up.load(); ++g) {
                int expect = 0;
                if (claim[g].compare_exchange_strong(expect, 1)) {
                    ops.touch(g);
                    touched[g].store(1);
                }
            }
        });
    // the calling thread: watch the first watermark move, then wait for the kernel
    int kstate = 0;
    {
        const double t0 = now();
        bool landed = false;
        while ((kstate = ops.kernel_state()) == 0) {
            if (failed.load()) break;
            if (!landed) landed = ops.first_piece_landed();
            if (landed) break;
            if (now() - t0 > stall_ms) {
                give_up.store(true);
                ops.abort_kernel();
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    const int kw = kstate == 0 ? ops.kernel_wait() : (kstate < 0 ? kstate : PNX_OK);
    std::string kmsg = kw ? last_error_text() : "";
    if (times) times->t_kernel = now();
    kernel_done.store(true);
    for (auto &t : th) t.join();
    if (kw) return set_error(kw, "%s", kmsg.c_str());
    if (give_up.load()) {
        *stalled = true;
        return PNX_OK;
    }
    if (failed.load()) {
        std::lock_guard<std::mutex> lk(err_mu);
        return set_error(err_code, "%s", err_msg.c_str());
    }
    return PNX_OK;
}

}  // namespace pnx
