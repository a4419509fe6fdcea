// host_entropy.cpp -- the product's host-side entropy coder (CPU, C++17, multi-threaded over images).
//
// Bit-compatible with the bitstream the reference produces through CompressAI 1.2.4's
// `compressai.ans` (rans64: 64-bit state, 32-bit words, 16-bit probabilities, 4-bit bypass digits,
// symbols applied in reverse so that decoding runs forwards) -- call sites
// hyperprior_charm_dc_vic_model.py:68,84 and minnen20_charm_context_model.py:165,179-202 -- and with
// `_CXX.pmf_to_quantized_cdf` (hyperprior_dc_vic_model.py:66-68).  Algorithm: SURVEY.md App-B.
// Differences from a textbook port (this is the shipped path, tuned for throughput):
//   * the encoder never materialises a symbol stack: it walks the symbols backwards and expands the
//     (rare) bypass escape of one symbol at a time;
//   * the decoder resolves cum-freq -> symbol through a 512-bucket jump table per CDF followed by a
//     short linear scan instead of a linear search from 0;
//   * streams (one per image / latent) are independent and are coded in parallel on host threads.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <vector>

#include "common.h"

namespace {
constexpr uint32_t kPrecision = 16;
constexpr uint32_t kBypassBits = 4;
constexpr uint32_t kBypassMax = 15;
constexpr uint64_t kL = 1ull << 31;
constexpr int kBuckets = 512;
constexpr int kBucketShift = 16 - 9;
}  // namespace

struct dcvic_cdf_tables {
    int n_cdf = 0, stride = 0;
    std::vector<int32_t> cdf;      // [n_cdf][stride]
    std::vector<int32_t> size;     // entries used (pmf_length + 2)
    std::vector<int32_t> offset;
    std::vector<uint16_t> jump;    // [n_cdf][kBuckets]: largest s with cdf[s] <= bucket start
};

struct dcvic_rans_decoder {
    std::vector<uint32_t> words;
    size_t pos = 0;
    uint64_t x = 0;
    bool bad = false;
};

extern "C" int dcvic_pmf_to_quantized_cdf_host(const float* pmf, int n, int32_t* cdf_out) {
    DCVIC_CHECK_ARG(pmf && cdf_out && n > 0, "pmf_to_quantized_cdf: bad argument");
    std::vector<uint32_t> cdf((size_t)n + 1);
    cdf[0] = 0;
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) {
        DCVIC_CHECK_ARG(pmf[i] >= 0.f && std::isfinite(pmf[i]), "pmf_to_quantized_cdf: invalid pmf[%d]", i);
        cdf[i + 1] = (uint32_t)std::round(pmf[i] * (float)(1 << kPrecision));
        total += cdf[i + 1];
    }
    DCVIC_CHECK_ARG(total != 0, "pmf_to_quantized_cdf: all-zero pmf");
    uint32_t run = 0;
    for (int i = 0; i <= n; ++i) {
        run += (uint32_t)((((uint64_t)1 << kPrecision) * cdf[i]) / total);
        cdf[i] = run;
    }
    cdf[n] = 1u << kPrecision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best_freq = ~0u;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best_freq) { best_freq = f; best = j; }
        }
        DCVIC_CHECK_ARG(best >= 0, "pmf_to_quantized_cdf: cannot steal a frequency");
        if (best < i) for (int j = best + 1; j <= i; ++j) cdf[j]--;
        else for (int j = i + 1; j <= best; ++j) cdf[j]++;
    }
    for (int i = 0; i <= n; ++i) cdf_out[i] = (int32_t)cdf[i];
    return DCVIC_OK;
}

extern "C" dcvic_cdf_tables* dcvic_tables_create_host(const int32_t* cdfs, int n_cdf, int stride, const int32_t* sizes,
                                                      const int32_t* offsets) {
    if (!cdfs || !sizes || !offsets || n_cdf <= 0 || stride < 3) { dcvic_set_error("tables_create: bad argument"); return nullptr; }
    auto* t = new dcvic_cdf_tables();
    t->n_cdf = n_cdf; t->stride = stride;
    t->cdf.assign(cdfs, cdfs + (size_t)n_cdf * stride);
    t->size.assign(sizes, sizes + n_cdf);
    t->offset.assign(offsets, offsets + n_cdf);
    t->jump.assign((size_t)n_cdf * kBuckets, 0);
    for (int c = 0; c < n_cdf; ++c) {
        const int32_t* cdf = &t->cdf[(size_t)c * stride];
        const int nsym = sizes[c] - 1;  // symbols 0..nsym-1 (the last one is the escape)
        if (sizes[c] < 2 || sizes[c] > stride || cdf[0] != 0 || cdf[nsym] != (1 << kPrecision)) {
            dcvic_set_error("tables_create: cdf %d malformed", c);
            delete t;
            return nullptr;
        }
        int s = 0;
        for (int b = 0; b < kBuckets; ++b) {
            const int32_t lo = b << kBucketShift;
            while (s + 1 < nsym && cdf[s + 1] <= lo) ++s;
            t->jump[(size_t)c * kBuckets + b] = (uint16_t)s;
        }
    }
    return t;
}

extern "C" void dcvic_tables_destroy_host(dcvic_cdf_tables* t) { delete t; }

namespace {

struct Emit { uint32_t start, range; bool bypass; };

// Encode one stream; returns bytes written or <0.
long long encode_one(const dcvic_cdf_tables& T, const int32_t* sym, const int32_t* idx, long long n, uint8_t* out, long long cap) {
    std::vector<uint32_t> buf((size_t)n * 2 + 16);
    uint32_t* const end = buf.data() + buf.size();
    uint32_t* ptr = end;
    uint64_t x = kL;
    Emit em[48];
    for (long long i = n - 1; i >= 0; --i) {
        const int32_t ci = idx[i];
        if (ci < 0 || ci >= T.n_cdf) return DCVIC_EINVAL;
        const int32_t* cdf = &T.cdf[(size_t)ci * T.stride];
        const int32_t max_value = T.size[ci] - 2;
        int32_t value = sym[i] - T.offset[ci];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * (int64_t)value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * ((int64_t)value - max_value)); value = max_value; }
        int ne = 0;
        em[ne++] = {(uint32_t)cdf[value], (uint32_t)(cdf[value + 1] - cdf[value]), false};
        if (value == max_value) {
            int32_t nb = 0;
            while ((raw >> (nb * kBypassBits)) != 0) ++nb;
            int32_t v = nb;
            while (v >= (int32_t)kBypassMax) { em[ne++] = {kBypassMax, 0, true}; v -= kBypassMax; }
            em[ne++] = {(uint32_t)v, 0, true};
            for (int32_t j = 0; j < nb; ++j) em[ne++] = {(raw >> (j * kBypassBits)) & kBypassMax, 0, true};
        }
        // capacity: <= 56 bits (16 + 10 bypass nibbles) per symbol, reserved above as 2 words per symbol
        for (int e = ne - 1; e >= 0; --e) {
            if (!em[e].bypass) {
                const uint32_t freq = em[e].range;
                const uint64_t x_max = ((kL >> kPrecision) << 32) * freq;
                if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
                x = ((x / freq) << kPrecision) + (x % freq) + em[e].start;
            } else {
                const uint64_t x_max = ((kL >> 16) << 32) * (uint64_t)(1u << (16 - kBypassBits));
                if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
                x = (x << kBypassBits) | em[e].start;
            }
        }
    }
    ptr -= 2;
    ptr[0] = (uint32_t)x;
    ptr[1] = (uint32_t)(x >> 32);
    const long long nbytes = (long long)(end - ptr) * 4;
    if (nbytes > cap) return DCVIC_ENOSPACE;
    memcpy(out, ptr, (size_t)nbytes);
    return nbytes;
}

inline uint32_t dec_bits(dcvic_rans_decoder& d, uint32_t nbits) {
    uint64_t x = d.x;
    const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < kL) {
        if (d.pos < d.words.size()) x = (x << 32) | d.words[d.pos++];
        else d.bad = true;
    }
    d.x = x;
    return val;
}

int decode_one(const dcvic_cdf_tables& T, dcvic_rans_decoder& d, const int32_t* idx, long long n, int32_t* out) {
    for (long long i = 0; i < n; ++i) {
        const int32_t ci = idx[i];
        if (ci < 0 || ci >= T.n_cdf) return DCVIC_EINVAL;
        const int32_t* cdf = &T.cdf[(size_t)ci * T.stride];
        const int32_t max_value = T.size[ci] - 2;
        const uint32_t cum = (uint32_t)(d.x & ((1u << kPrecision) - 1));
        int32_t s = T.jump[(size_t)ci * kBuckets + (cum >> kBucketShift)];
        while ((uint32_t)cdf[s + 1] <= cum) ++s;
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        uint64_t x = d.x;
        x = freq * (x >> kPrecision) + (x & ((1ull << kPrecision) - 1)) - start;
        if (x < kL) {
            if (d.pos < d.words.size()) x = (x << 32) | d.words[d.pos++];
            else d.bad = true;
        }
        d.x = x;
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_bits(d, kBypassBits);
            int32_t nb = val;
            while (val == (int32_t)kBypassMax && !d.bad) { val = (int32_t)dec_bits(d, kBypassBits); nb += val; }
            if (nb > 8) return DCVIC_ECORRUPT;
            int32_t raw = 0;
            for (int32_t j = 0; j < nb; ++j) { val = (int32_t)dec_bits(d, kBypassBits); raw |= val << (j * kBypassBits); }
            value = raw >> 1;
            if (raw & 1) value = -value - 1; else value += max_value;
        }
        out[i] = value + T.offset[ci];
    }
    return DCVIC_OK;
}

// Persistent worker pool (one per process, grown on demand).  The decoder calls parallel_for six times per batch for ~0.3 ms of work
// each: spawning std::threads per call (round 2) cost as much as the decode itself beyond 16 threads.  Tasks are handed out through an
// atomic counter; the calling thread works too; at most `threads` threads touch a call.  Calls are serialised by `run_mutex` (two Python
// threads coding at once take turns).
class WorkerPool {
public:
    static WorkerPool& get() { static WorkerPool p; return p; }
    template <typename F>
    void run(int n, int threads, F&& f) {
        std::lock_guard<std::mutex> serial(run_mutex);
        ensure(threads - 1);
        std::function<void(int)> fn = [&](int i) { f(i); };
        {
            std::lock_guard<std::mutex> lk(m);
            task = &fn; n_tasks = n; next.store(0); done = 0; helpers = threads - 1; ++generation;
        }
        cv_work.notify_all();
        work(fn, n);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return done == n_tasks && busy == 0; });
        task = nullptr;
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; ++generation; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
private:
    void work(const std::function<void(int)>& fn, int n) {
        int finished = 0;
        for (int i; (i = next.fetch_add(1)) < n;) { fn(i); ++finished; }
        if (finished) { std::lock_guard<std::mutex> lk(m); done += finished; if (done == n_tasks) cv_done.notify_all(); }
    }
    void ensure(int count) {
        while ((int)workers.size() < count) {
            const int id = (int)workers.size();
            workers.emplace_back([this, id]() {
                unsigned long long seen = 0;
                for (;;) {
                    const std::function<void(int)>* fn; int n;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_work.wait(lk, [&] { return stop || (generation != seen && id < helpers && task); });
                        if (stop) return;
                        seen = generation; fn = task; n = n_tasks; ++busy;
                    }
                    work(*fn, n);
                    { std::lock_guard<std::mutex> lk(m); --busy; if (done == n_tasks && busy == 0) cv_done.notify_all(); }
                }
            });
        }
    }
    std::vector<std::thread> workers;
    std::mutex m, run_mutex;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)>* task = nullptr;
    std::atomic<int> next{0};
    int n_tasks = 0, done = 0, helpers = 0, busy = 0;
    unsigned long long generation = 0;
    bool stop = false;
};

template <typename F>
void parallel_for(int n, int threads, F&& f) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (threads == 1) { for (int i = 0; i < n; ++i) f(i); return; }
    WorkerPool::get().run(n, threads, f);
}

}  // namespace

extern "C" int dcvic_rans_encode_batch_host(const dcvic_cdf_tables* T, const int32_t* symbols, const int32_t* indexes,
                                            int n_streams, long long n_sym, uint8_t* out, long long out_cap, long long* out_len,
                                            int threads) {
    DCVIC_CHECK_ARG(T && symbols && indexes && out && out_len && n_streams > 0 && n_sym >= 0 && out_cap >= 8, "rans_encode: bad argument");
    std::vector<long long> rc((size_t)n_streams, 0);
    parallel_for(n_streams, threads, [&](int i) {
        rc[i] = encode_one(*T, symbols + (long long)i * n_sym, indexes + (long long)i * n_sym, n_sym, out + (long long)i * out_cap, out_cap);
    });
    for (int i = 0; i < n_streams; ++i) {
        if (rc[i] < 0) {
            dcvic_set_error("rans_encode: stream %d failed (%lld)", i, rc[i]);
            return rc[i] == DCVIC_ENOSPACE ? DCVIC_ENOSPACE : DCVIC_EINVAL;
        }
        out_len[i] = rc[i];
    }
    return DCVIC_OK;
}

extern "C" dcvic_rans_decoder* dcvic_rans_decoder_create_host(const uint8_t* stream, long long nbytes) {
    if (!stream || nbytes < 8 || (nbytes & 3)) { dcvic_set_error("rans_decoder_create: stream of %lld bytes", nbytes); return nullptr; }
    auto* d = new dcvic_rans_decoder();
    d->words.resize((size_t)nbytes / 4);
    memcpy(d->words.data(), stream, (size_t)nbytes);
    d->x = (uint64_t)d->words[0] | ((uint64_t)d->words[1] << 32);
    d->pos = 2;
    return d;
}

extern "C" void dcvic_rans_decoder_destroy_host(dcvic_rans_decoder* d) { delete d; }

extern "C" int dcvic_rans_decode_batch_host(const dcvic_cdf_tables* T, dcvic_rans_decoder* const* decoders, const int32_t* indexes,
                                            int n_streams, long long n_sym, int32_t* symbols, int threads) {
    DCVIC_CHECK_ARG(T && decoders && indexes && symbols && n_streams > 0 && n_sym >= 0, "rans_decode: bad argument");
    std::vector<int> rc((size_t)n_streams, 0);
    parallel_for(n_streams, threads, [&](int i) {
        if (!decoders[i]) { rc[i] = DCVIC_EINVAL; return; }
        rc[i] = decode_one(*T, *decoders[i], indexes + (long long)i * n_sym, n_sym, symbols + (long long)i * n_sym);
        if (rc[i] == DCVIC_OK && decoders[i]->bad) rc[i] = DCVIC_ECORRUPT;
    });
    for (int i = 0; i < n_streams; ++i)
        if (rc[i] != DCVIC_OK) { dcvic_set_error("rans_decode: stream %d failed (%d)", i, rc[i]); return rc[i]; }
    return DCVIC_OK;
}
