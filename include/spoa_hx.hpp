// spoa_hx.hpp — the per-edge POA operator of haslr_assemble behind the reference's own operator API.
//
// The reference computes every gap consensus with five calls of rvaser/spoa 1.1.3 (Assemble.cpp:499-554):
//     auto alignment_engine = spoa::createAlignmentEngine(static_cast<spoa::AlignmentType>(1), 5, -4, -8);   // :499
//     auto graph = spoa::createGraph();                                                                         // :500
//     for every supporting sub-sequence, in stored order:
//         auto alignment = alignment_engine->align_sequence_with_graph(seq, graph);                            // :539
//         graph->add_alignment(alignment, seq);                                                                 // :540
//     std::string consensus = graph->generate_consensus();                                                      // :554
// This header declares exactly those symbols (namespace spoa, the same signatures and ownership: unique_ptr
// engine + graph per edge, strings by const reference, consensus by value) over the C-ABI of haslr_hip.h, so that
// the reference's Assemble.cpp compiles against it unchanged (`#include "spoa_hx.hpp"` in place of "spoa.hpp",
// link -lhaslr_hip instead of libspoa.a) and its consensus runs on the MI355X.
//
// How it maps: a partial-order graph lives on the device only while it is being built, so the calls are recorded and
// the work happens in generate_consensus(): align_sequence_with_graph() returns a token (an Alignment holding one
// (-1, ticket) pair, not a list of node/position pairs), add_alignment() appends the sequence that goes with a token,
// generate_consensus() sends the recorded sequences, in order, through hx_poa_sequences (global alignment, linear gap,
// the engine's three scores, unit weights) and returns what spoa's generate_consensus returns for them.
// What that supports is the reference's call pattern and nothing wider: AlignmentType::kNW only (createAlignmentEngine
// throws std::invalid_argument otherwise), weight 1 only, every alignment added to the graph it was computed against,
// in the order it was computed. spoa 1.1.3 exits on invalid input; this header throws std::runtime_error with the
// library's message instead (there is no CPU fallback: without a HIP device every consensus fails loudly).
//
// Threads: the reference calls from gopt.num_threads pthreads, each with its own engine and graph (asm_cal_cns_seq_MT, Assemble.cpp:562-605).
// All of them share one device context here (created on first use, device HASLR_DEVICE or 0), and their generate_consensus() calls are
// FLAT-COMBINED: a caller queues its sequence set; the first one in becomes the submitter, waits HASLR_SPOA_BATCH_US microseconds (default
// 200) or until HASLR_SPOA_BATCH sets (default 256) are queued - only while it has company: a lone caller (-t 1) submits at once - and sends
// everything queued through ONE hx_poa_sequences call; callers that arrive while a call is on the device form the next batch. A set that makes the
// shared call fail is isolated (every set of that call again, on its own): only its caller gets the exception. With -t 64 the reference's own thread fan-out therefore puts ~64 edges into
// every launch instead of one. spoa::hx::consensus_batch() below is the entry to use from new code: all edges in one call (that is what
// haslr_amd's own pipeline does through hx_poa_batch). spoa::hx::stats() tells how many device calls served how many sets.
//
// This is product code. It is never used to build oracle/_ref (a reference build must not be made with stand-in
// headers): tests/test_spoa_header.py compiles a small caller written against the five symbols, nothing else.
#ifndef HASLR_SPOA_HX_HPP
#define HASLR_SPOA_HX_HPP
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "haslr_hip.h"

namespace spoa {

enum class AlignmentType { kSW, kNW, kOV };   // values 0, 1, 2 as in spoa 1.1.3 (the reference passes 1)
using Alignment = std::vector<std::pair<std::int32_t, std::int32_t>>;

namespace hx {

struct Device {
    hx_ctx* ctx = nullptr;
    std::mutex mu;                    // guards ctx and every call into it
    // flat combining of concurrent generate_consensus() calls
    struct Request { const std::vector<std::string>* seqs; std::int8_t m, n, g; std::string result, error; bool done, answered; };   // answered: result or error is final (an empty consensus is a result)
    std::mutex qmu;
    std::condition_variable qcv;
    std::vector<Request*> queue;
    bool submitting = false;          // a submitter is collecting or has a batch on the device
    bool company = false;             // the previous batch had, or its device call saw, more than one caller: the next submitter waits its window out
    std::size_t arrivals = 0;         // requests queued since the current batch was taken
    std::uint64_t calls = 0, sets = 0;
    // (no destructor: this object is destroyed during static destruction, possibly after the HIP runtime has torn itself down -
    //  the context and its device memory are left to the end of the process; spoa::hx::shutdown() releases them explicitly)
};
inline Device& device() {
    static Device d;
    return d;
}
inline hx_ctx* context_locked(Device& d) {   // call with d.mu held
    if (!d.ctx) {
        // before HIP initialises: the POA launch classes overlap on separate hardware queues (include/haslr_hip.h; an application's own setting wins)
        setenv("GPU_MAX_HW_QUEUES", "8", 0);
        const char* dev = std::getenv("HASLR_DEVICE");
        if (hx_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &d.ctx) != 0) throw std::runtime_error(std::string("spoa_hx: ") + hx_last_error());
    }
    return d.ctx;
}
// releases the shared device context (optional: call it when no Graph is in use any more, before main returns)
inline void shutdown() {
    Device& d = device();
    std::lock_guard<std::mutex> lock(d.mu);
    if (d.ctx) { hx_ctx_destroy(d.ctx); d.ctx = nullptr; }
}
struct Stats { std::uint64_t device_calls, sets; };
inline Stats stats() {
    Device& d = device();
    std::lock_guard<std::mutex> lock(d.qmu);
    return Stats{d.calls, d.sets};
}

// consensus of every set of sequences (set = the sub-sequences of one edge in alignment order) in ONE device call
inline std::vector<std::string> consensus_batch(const std::vector<const std::vector<std::string>*>& sets, std::int8_t m = 5, std::int8_t n = -4, std::int8_t g = -8) {
    std::vector<std::uint64_t> set_off{0}, seq_off{0};
    std::string bases;
    for (const auto* st : sets) {
        for (const auto& s : *st) { bases += s; seq_off.push_back(bases.size()); }
        set_off.push_back(seq_off.size() - 1);
    }
    const hx_poa_params pp{m, n, g};
    hx_cns_out out;
    Device& d = device();
    std::lock_guard<std::mutex> lock(d.mu);
    hx_ctx* ctx = context_locked(d);
    if (hx_poa_sequences(ctx, (std::uint32_t)sets.size(), set_off.data(), seq_off.data(), bases.c_str(), &pp, &out) != 0)
        throw std::runtime_error(std::string("spoa_hx: ") + hx_last_error());
    std::vector<std::string> res(sets.size());
    for (std::size_t i = 0; i < sets.size(); i++) res[i].assign(out.cns + out.cns_off[i], out.cns + out.cns_off[i + 1]);
    hx_free_cns(ctx, &out);
    return res;
}
inline std::vector<std::string> consensus_batch(const std::vector<std::vector<std::string>>& sets, std::int8_t m = 5, std::int8_t n = -4, std::int8_t g = -8) {
    std::vector<const std::vector<std::string>*> p;
    for (const auto& st : sets) p.push_back(&st);
    return consensus_batch(p, m, n, g);
}

// one set on behalf of one caller thread, combined with whatever other threads have queued (see "Threads" above)
inline std::string consensus_combined(const std::vector<std::string>& seqs, std::int8_t m, std::int8_t n, std::int8_t g) {
    Device& d = device();
    static const long window_us = std::getenv("HASLR_SPOA_BATCH_US") ? std::atol(std::getenv("HASLR_SPOA_BATCH_US")) : 200;
    static const std::size_t batch_max = std::getenv("HASLR_SPOA_BATCH") ? (std::size_t)std::max(1L, std::atol(std::getenv("HASLR_SPOA_BATCH"))) : 256;
    Device::Request me{&seqs, m, n, g, std::string(), std::string(), false, false};
    std::unique_lock<std::mutex> lk(d.qmu);
    d.queue.push_back(&me);
    d.arrivals++;
    d.qcv.notify_all();                                        // (a submitter that is collecting counts the queue)
    while (!me.done) {
        if (d.submitting) { d.qcv.wait(lk); continue; }         // somebody else submits: my request rides along, or waits for the next batch
        d.submitting = true;                                    // I submit: collect for the window, then take everything queued
        std::vector<Device::Request*> batch;
        // Whatever goes wrong from here on (an allocation that throws as well), `submitting` is reset, the requests taken so far are answered (with
        // the error) and the waiters are woken: no thread may be left waiting on a submitter that is gone.
        try {
            // The window is for company: a lone caller (the reference with -t 1, or the only thread that still has edges) would sit through it on
            // every call for nothing. It is waited out only when somebody else has been seen since the previous batch closed (d.company), or is
            // queued right now.
            if (d.company || d.queue.size() > 1) {
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
                while (d.queue.size() < batch_max && d.qcv.wait_until(lk, until) != std::cv_status::timeout) { }
            }
            batch.swap(d.queue);
            if (batch.size() > batch_max) { d.queue.assign(batch.begin() + (std::ptrdiff_t)batch_max, batch.end()); batch.resize(batch_max); }
            d.arrivals = d.queue.size();                           // arrivals from here on = company for the next submitter
            lk.unlock();
            // one device call per score triple in the batch (the reference uses one triple)
            std::vector<char> served(batch.size(), 0);
            std::uint64_t calls = 0;
            for (std::size_t i = 0; i < batch.size(); i++) {
                if (served[i]) continue;
                std::vector<std::size_t> idx;
                std::vector<const std::vector<std::string>*> sets;
                for (std::size_t j = i; j < batch.size(); j++)
                    if (!served[j] && batch[j]->m == batch[i]->m && batch[j]->n == batch[i]->n && batch[j]->g == batch[i]->g) { idx.push_back(j); sets.push_back(batch[j]->seqs); served[j] = 1; }
                try {
                    std::vector<std::string> res = consensus_batch(sets, batch[i]->m, batch[i]->n, batch[i]->g);
                    for (std::size_t q = 0; q < idx.size(); q++) { batch[idx[q]]->result.swap(res[q]); batch[idx[q]]->answered = true; }
                    calls++;
                } catch (const std::exception& e) {
                    // one bad set must not fail the callers that happened to share its launch: every set of the group again, on its own - until two in a
                    // row have failed the way the whole group did (a device fault is sticky: the other 250 sets would fail one call at a time)
                    const std::string group_error = e.what();
                    if (idx.size() == 1) { batch[idx[0]]->error = group_error; batch[idx[0]]->answered = true; }
                    else {
                        int same = 0;
                        for (std::size_t q = 0; q < idx.size(); q++) {
                            Device::Request* r = batch[idx[q]];
                            if (same >= 2) { r->error = group_error; r->answered = true; continue; }
                            try { r->result = consensus_batch(std::vector<const std::vector<std::string>*>{sets[q]}, batch[i]->m, batch[i]->n, batch[i]->g)[0]; same = 0; }
                            catch (const std::exception& e1) { r->error = e1.what(); same = r->error == group_error ? same + 1 : 0; }
                            r->answered = true;
                            calls++;
                        }
                    }
                    calls++;
                }
            }
            lk.lock();
            d.calls += calls; d.sets += batch.size();
        } catch (...) {
            if (!lk.owns_lock()) lk.lock();
            // requests of the batch that have no answer yet get the error (an EMPTY consensus that was served is an answer and stays)
            for (Device::Request* r : batch) if (!r->answered) { r->error = "spoa_hx: the submitting thread failed before the device call (out of memory?)"; r->answered = true; }
            if (batch.empty()) {   // failed while collecting: nothing was taken - this caller gets the error and leaves the queue
                me.error = "spoa_hx: the submitting thread failed while collecting a batch (out of memory?)";
                for (std::size_t q = 0; q < d.queue.size(); q++) if (d.queue[q] == &me) { d.queue.erase(d.queue.begin() + (std::ptrdiff_t)q); break; }
                me.done = true;
            }
            // (a batch cut at batch_max may have left this caller's own request in the queue: it stays there for the next submitter - possibly this thread)
        }
        for (Device::Request* r : batch) r->done = true;
        d.company = batch.size() > 1 || d.arrivals > 0;         // did this batch have, or did its device call see, anybody else?
        d.submitting = false;
        d.qcv.notify_all();
    }
    lk.unlock();
    if (!me.error.empty()) throw std::runtime_error(me.error);
    return me.result;
}

}  // namespace hx

class Graph {
public:
    // spoa::Graph::add_alignment(alignment, sequence, weight = 1)
    void add_alignment(const Alignment& alignment, const std::string& sequence, std::uint32_t weight = 1) {
        if (weight != 1) throw std::invalid_argument("spoa_hx: only unit weights are supported (the reference uses the default)");
        if (alignment.size() != 1 || alignment[0].first != -1 || (std::uint32_t)alignment[0].second != ticket_)
            throw std::invalid_argument("spoa_hx: add_alignment needs the alignment that align_sequence_with_graph last returned for this graph");
        ticket_++;
        if (!sequence.empty()) sequences_.push_back(sequence);   // spoa ignores an empty sequence (the reference never passes one, Assemble.cpp:537)
    }
    // spoa::Graph::generate_consensus()
    std::string generate_consensus() {
        if (sequences_.empty()) return std::string();
        return hx::consensus_combined(sequences_, m_, n_, g_);
    }

private:
    friend class AlignmentEngine;
    std::vector<std::string> sequences_;
    std::uint32_t ticket_ = 0;
    std::int8_t m_ = 5, n_ = -4, g_ = -8;
};

class AlignmentEngine {
public:
    // spoa::AlignmentEngine::align_sequence_with_graph(sequence, graph): a token for add_alignment (see the header comment)
    Alignment align_sequence_with_graph(const std::string& /*sequence*/, const std::unique_ptr<Graph>& graph) {
        graph->m_ = m_; graph->n_ = n_; graph->g_ = g_;
        return Alignment{{-1, (std::int32_t)graph->ticket_}};
    }

private:
    friend std::unique_ptr<AlignmentEngine> createAlignmentEngine(AlignmentType, std::int8_t, std::int8_t, std::int8_t);
    AlignmentEngine(std::int8_t m, std::int8_t n, std::int8_t g) : m_(m), n_(n), g_(g) {}
    std::int8_t m_, n_, g_;
};

// spoa::createAlignmentEngine(type, match, mismatch, gap) — linear gap penalties, as spoa 1.1.3 has them
inline std::unique_ptr<AlignmentEngine> createAlignmentEngine(AlignmentType type, std::int8_t m, std::int8_t n, std::int8_t g) {
    if (type != AlignmentType::kNW) throw std::invalid_argument("spoa_hx: only AlignmentType::kNW (global alignment, what the reference uses) is implemented");
    if (g >= 0) throw std::invalid_argument("spoa_hx: the gap penalty must be negative");
    return std::unique_ptr<AlignmentEngine>(new AlignmentEngine(m, n, g));
}
inline std::unique_ptr<Graph> createGraph() { return std::unique_ptr<Graph>(new Graph()); }

}  // namespace spoa
#endif
