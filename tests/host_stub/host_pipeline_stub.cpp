// host_pipeline_stub.cpp -- pyneapple_amd/csrc/pnx_host_pipeline.hpp (the thread / flag orchestration of PNX_MEM_HOST calls)
// built for the CPU against a stub device, to be run under ThreadSanitizer and AddressSanitizer (tests/test_host_sanitizers.py;
// GPU sanitizers are not available on the pool, and the orchestration is host code anyway).
//
// The stub device: a stream is a worker thread that executes closures in order (copies are memcpy, kernels are loops with a
// short sleep), an event is a flag raised by a closure, the persistent streamed kernel is a thread that polls the upload
// watermark / the abort word and raises per-granule flags -- the same protocol as curvefit_kernel<..., STREAM = true>
// (pnx_curvefit_kernel.hpp: refill / publish).  Test infrastructure: nothing here is part of the product.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "pnx_host_pipeline.hpp"

namespace pnx {
static thread_local char g_err[512] = "";
int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    char tmp[sizeof(g_err)];
    vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    memcpy(g_err, tmp, sizeof(g_err));
    return code;
}
const char *last_error_text() { return g_err; }
}  // namespace pnx
using namespace pnx;

// ---- stub runtime -----------------------------------------------------------------------------------------------
struct StubStream {
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    StubStream() {
        worker = std::thread([this]() {
            for (;;) {
                std::function<void()> f;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    f = std::move(q.front());
                    q.pop_front();
                    busy = true;
                }
                f();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    busy = false;
                }
                idle.notify_all();
            }
        });
    }
    ~StubStream() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        worker.join();
    }
    void push(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(f));
        }
        cv.notify_all();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(mu);
        idle.wait(lk, [&] { return q.empty() && !busy; });
    }
};
struct StubEvent {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
};
static std::atomic<int> g_live_streams(0), g_live_events(0);
struct StubBackend {
    typedef StubStream *stream_t;
    typedef StubEvent *event_t;
    static bool bind_device(int) { return true; }
    static bool stream_create(stream_t *s, bool) {
        *s = new StubStream();
        g_live_streams++;
        return true;
    }
    static void stream_destroy(stream_t s) {
        delete s;
        g_live_streams--;
    }
    static bool stream_sync(stream_t s) {
        if (s) s->sync();
        return true;
    }
    static bool event_create(event_t *e) {
        *e = new StubEvent();
        g_live_events++;
        return true;
    }
    static void event_destroy(event_t e) {
        delete e;
        g_live_events--;
    }
    static bool event_record(event_t e, stream_t s) {
        s->push([e]() {
            {
                std::lock_guard<std::mutex> lk(e->mu);
                e->done = true;
            }
            e->cv.notify_all();
        });
        return true;
    }
    static bool event_sync(event_t e) {
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv.wait(lk, [&] { return e->done; });
        return true;
    }
};

static int g_failures = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "CHECK failed: %s -- ", #cond); \
            fprintf(stderr, __VA_ARGS__);                 \
            fprintf(stderr, "\n");                        \
            ++g_failures;                                 \
        }                                                 \
    } while (0)

static double model(double v) { return 3.0 * v + 1.0; }

// ---- the chunk ring -----------------------------------------------------------------------------------------------
// fail_stage: 0 none, 1 h2d, 2 launch, 3 d2h (at chunk fail_at)
static void ring_case(int n_vox, int chunk, int n_slots, int k_streams, int touchers, int fail_stage, int fail_at) {
    std::vector<double> in(n_vox), out(n_vox, -7.0);
    for (int i = 0; i < n_vox; ++i) in[i] = 0.5 * i;
    const int n_chunks = (n_vox + chunk - 1) / chunk;
    const int slots = n_chunks < n_slots ? n_chunks : n_slots;
    std::vector<std::vector<double>> dev_in(slots, std::vector<double>(chunk)), dev_out(slots, std::vector<double>(chunk));
    auto span = [&](int k, int &v0, int &c) {
        v0 = k * chunk;
        c = std::min(chunk, n_vox - v0);
    };
    PipeOpsT<StubStream *> ops;
    StubStream user;  // the caller's stream (single-chunk calls run on it)
    ops.h2d = [&](int k, int slot, StubStream *st) -> int {
        if (fail_stage == 1 && k == fail_at) return set_error(PNX_ERR_HIP, "injected H2D failure at chunk %d", k);
        int v0, c;
        span(k, v0, c);
        st->push([&, slot, v0, c]() { memcpy(dev_in[slot].data(), in.data() + v0, sizeof(double) * c); });
        return PNX_OK;
    };
    ops.launch = [&](int k, int slot, StubStream *st) -> int {
        if (fail_stage == 2 && k == fail_at) return set_error(PNX_ERR_HIP, "injected launch failure at chunk %d", k);
        int v0, c;
        span(k, v0, c);
        st->push([&, slot, c]() {
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            for (int i = 0; i < c; ++i) dev_out[slot][i] = model(dev_in[slot][i]);
        });
        return PNX_OK;
    };
    ops.d2h = [&](int k, int slot, StubStream *st) -> int {
        if (fail_stage == 3 && k == fail_at) return set_error(PNX_ERR_HIP, "injected D2H failure at chunk %d", k);
        int v0, c;
        span(k, v0, c);
        st->push([&, slot, v0, c]() { memcpy(out.data() + v0, dev_out[slot].data(), sizeof(double) * c); });
        return PNX_OK;
    };
    ops.touch = [&](int k) {  // first-touch of the chunk's result range: plain stores, as touch_pages does
        int v0, c;
        span(k, v0, c);
        for (int i = 0; i < c; i += 512) out[v0 + i] = 0.0;
    };
    const int rc = run_pipeline_t<StubBackend>(n_chunks, slots, k_streams, touchers, 0, &user, ops, false);
    if (fail_stage == 0 || fail_at >= n_chunks) {
        CHECK(rc == PNX_OK, "ring rc=%d (%s)", rc, last_error_text());
        int bad = 0;
        for (int i = 0; i < n_vox; ++i) bad += out[i] != model(in[i]);
        CHECK(bad == 0, "ring: %d wrong results (n_vox=%d chunk=%d slots=%d k=%d touchers=%d)", bad, n_vox, chunk, n_slots, k_streams, touchers);
    } else {
        CHECK(rc == PNX_ERR_HIP, "ring with an injected failure returned %d", rc);
        CHECK(strstr(last_error_text(), "injected") != nullptr, "the helper's message did not reach the caller: '%s'", last_error_text());
    }
}

// ---- one streamed kernel ----------------------------------------------------------------------------------------
struct FakeDevice {
    std::atomic<unsigned long long> ready{0};  // StreamCtl::ready
    std::atomic<unsigned int> abort_word{0};   // host_flags[n_granules]
    std::vector<std::atomic<unsigned int>> flags;
    std::atomic<int> timed_out{0}, kernel_ended{0};
    std::vector<double> in, out;
};
// fail: 0 none, 1 upload of piece fail_at, 2 download of granule fail_at; delay_ms: the IN thread sleeps first (a stalled upload)
static void streamed_case(int n_vox, int piece, int gran, int n_out, int touchers, int fail, int fail_at, int delay_ms, double stall_ms,
                          bool expect_stall) {
    const int n_in = (n_vox + piece - 1) / piece, n_gran = (n_vox + gran - 1) / gran;
    std::vector<double> src(n_vox), res(n_vox, -7.0);
    for (int i = 0; i < n_vox; ++i) src[i] = 0.25 * i;
    FakeDevice D;
    D.flags = std::vector<std::atomic<unsigned int>>(n_gran);
    for (auto &f : D.flags) f.store(0);
    D.in.assign(n_vox, 0.0);
    D.out.assign(n_vox, 0.0);
    StubStream s_in;
    std::vector<StubStream *> s_out;
    for (int i = 0; i < n_out; ++i) s_out.push_back(new StubStream());
    const auto t0 = std::chrono::steady_clock::now();
    auto now = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    // the persistent kernel, launched before the first byte is uploaded: granule by granule, waits for the watermark (bounded),
    // leaves on the abort word
    std::thread kernel([&]() {
        for (int g = 0; g < n_gran; ++g) {
            const int v0 = g * gran, v1 = std::min(n_vox, v0 + gran);
            bool got = false;
            for (int spins = 0; spins < 400000; ++spins) {
                if (D.ready.load(std::memory_order_acquire) >= (unsigned long long)v1) {
                    got = true;
                    break;
                }
                if (D.abort_word.load(std::memory_order_relaxed)) break;
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            if (!got) {
                if (!D.abort_word.load()) D.timed_out.store(1);
                break;
            }
            for (int i = v0; i < v1; ++i) D.out[i] = model(D.in[i]);
            D.flags[g].store(1, std::memory_order_release);
        }
        D.kernel_ended.store(1, std::memory_order_release);
    });
    std::atomic<int> first_landed(0);
    StreamedOps ops;
    ops.bind_device = []() { return true; };
    ops.upload_piece = [&](int i) -> int {
        if (fail == 1 && i == fail_at) return set_error(PNX_ERR_HIP, "injected upload failure at piece %d", i);
        const int v0 = i * piece, c = std::min(piece, n_vox - v0);
        s_in.push([&, v0, c, i]() {
            memcpy(D.in.data() + v0, src.data() + v0, sizeof(double) * c);
            D.ready.store((unsigned long long)(v0 + c), std::memory_order_release);  // the watermark move behind the data
            if (i == 0) first_landed.store(1);
        });
        return PNX_OK;
    };
    ops.upload_sync = [&]() -> int {
        s_in.sync();
        return PNX_OK;
    };
    ops.first_piece_landed = [&]() { return first_landed.load() != 0; };
    ops.granule_ready = [&](int g) { return D.flags[g].load(std::memory_order_acquire) != 0; };
    ops.download = [&](int g, int ot) -> int {
        if (fail == 2 && g == fail_at) return set_error(PNX_ERR_HIP, "injected download failure at granule %d", g);
        const int v0 = g * gran, c = std::min(gran, n_vox - v0);
        s_out[ot]->push([&, v0, c]() { memcpy(res.data() + v0, D.out.data() + v0, sizeof(double) * c); });
        s_out[ot]->sync();
        return PNX_OK;
    };
    ops.touch = [&](int g) {
        const int v0 = g * gran, c = std::min(gran, n_vox - v0);
        for (int i = 0; i < c; i += 512) res[v0 + i] = 0.0;
    };
    ops.abort_kernel = [&]() { D.abort_word.store(1, std::memory_order_release); };
    ops.kernel_state = [&]() -> int { return D.kernel_ended.load(std::memory_order_acquire) ? 1 : 0; };
    ops.kernel_wait = [&]() -> int {
        while (!D.kernel_ended.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        return PNX_OK;
    };
    bool stalled = false;
    StreamedTimes times;
    const double t_start = now();
    const int rc = run_streamed(n_in, n_gran, n_out, touchers, stall_ms, delay_ms, ops, &stalled, &times, now);
    const double took = now() - t_start;
    kernel.join();
    s_in.sync();
    for (auto s : s_out) delete s;
    if (expect_stall) {
        CHECK(rc == PNX_OK && stalled, "stalled upload: rc=%d stalled=%d", rc, (int)stalled);
        CHECK(took < stall_ms + 150.0, "a stalled call took %.1f ms (stall limit %.0f ms): the give-up is not prompt", took, stall_ms);
        CHECK(D.timed_out.load() == 0, "the kernel ran into its own poll limit although the host gave up");
    } else if (fail) {
        CHECK(rc == PNX_ERR_HIP, "streamed call with an injected failure returned %d", rc);
        CHECK(strstr(last_error_text(), "injected") != nullptr, "the helper's message did not reach the caller: '%s'", last_error_text());
        CHECK(D.abort_word.load() == 1, "a helper failed and nobody told the kernel");
        CHECK(took < 2000.0, "a failing call took %.1f ms", took);
    } else {
        CHECK(rc == PNX_OK && !stalled, "streamed rc=%d stalled=%d (%s)", rc, (int)stalled, last_error_text());
        int bad = 0;
        for (int i = 0; i < n_vox; ++i) bad += res[i] != model(src[i]);
        CHECK(bad == 0, "streamed: %d wrong results (n_vox=%d piece=%d gran=%d n_out=%d touchers=%d)", bad, n_vox, piece, gran, n_out, touchers);
    }
}

int main() {
    // chunk ring: one chunk (no threads), two, many; slots 2-4; one or two kernel streams; 0-3 page-touch helpers
    ring_case(1000, 4096, 3, 2, 2, 0, 0);
    ring_case(5000, 2500, 3, 1, 0, 0, 0);
    for (int touchers = 0; touchers <= 3; ++touchers)
        for (int slots = 2; slots <= 4; ++slots) ring_case(40000 + 37 * touchers, 3000 + 11 * slots, slots, 1 + (slots & 1), touchers, 0, 0);
    // a failing stage at the first, a middle and the last chunk: the call returns the helper's error, nothing hangs
    for (int stage = 1; stage <= 3; ++stage)
        for (int at : {0, 4, 9}) ring_case(30000, 3000, 3, 2, 2, stage, at);
    // streamed call: ragged granules, pieces that do not line up with granules, 1-3 download threads, 0-3 helpers
    for (int n_out = 1; n_out <= 3; ++n_out)
        for (int touchers = 0; touchers <= 3; ++touchers) streamed_case(50000 + 13 * touchers, 4096 + 7 * n_out, 8192, n_out, touchers, 0, 0, 0, 2000.0, false);
    streamed_case(100, 4096, 8192, 2, 2, 0, 0, 0, 2000.0, false);  // one piece, one granule
    // failures: upload of the first / a later piece, download of the first / a later granule
    streamed_case(60000, 4096, 8192, 2, 2, 1, 0, 0, 2000.0, false);
    streamed_case(60000, 4096, 8192, 2, 2, 1, 5, 0, 2000.0, false);
    streamed_case(60000, 4096, 8192, 2, 2, 2, 0, 0, 2000.0, false);
    streamed_case(60000, 4096, 8192, 2, 2, 2, 4, 0, 2000.0, false);
    // a stalled upload: the host gives up after the stall limit, tells the kernel, and the call is over well before the stall ends
    streamed_case(60000, 4096, 8192, 2, 2, 0, 0, 3000, 40.0, true);
    streamed_case(60000, 4096, 8192, 1, 0, 0, 0, 3000, 20.0, true);
    CHECK(g_live_streams.load() == 0 && g_live_events.load() == 0, "leaked %d streams / %d events", g_live_streams.load(), g_live_events.load());
    if (g_failures) {
        fprintf(stderr, "host pipeline stub: %d check(s) failed\n", g_failures);
        return 1;
    }
    printf("host pipeline stub ok\n");
    return 0;
}
