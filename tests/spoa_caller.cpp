// Test program for include/spoa_hx.hpp: a caller written against the five spoa symbols the reference uses
// (Assemble.cpp:499,500,539,540,554), in the reference's call pattern — one engine + one graph per edge, sequences aligned and added
// one after the other, consensus at the end. Input: edges separated by blank lines, one sequence per line; output: one consensus per line.
// With --batch the same sets go through spoa::hx::consensus_batch in one call. With --threads N the edges are dealt to N threads, each in the
// reference's pattern with its own engine and graph (asm_cal_cns_seq_MT, Assemble.cpp:562-605); stderr then says how many device calls served them.
// --tolerant (with --threads): an edge whose consensus throws prints "ERROR <message>" in its place and the others go on (who gets the exception of a bad set?).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "spoa_hx.hpp"

int main(int argc, char** argv) {
    const bool batch = argc > 1 && !strcmp(argv[1], "--batch");
    std::vector<std::vector<std::string>> edges(1);
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) edges.emplace_back();
        else edges.back().push_back(line == "-" ? std::string() : line);
    }
    if (edges.back().empty()) edges.pop_back();
    try {
        if (batch) {
            for (const std::string& c : spoa::hx::consensus_batch(edges)) printf("%s\n", c.c_str());
            return 0;
        }
        const int nthreads = argc > 2 && !strcmp(argv[1], "--threads") ? std::max(1, atoi(argv[2])) : 1;
        const bool tolerant = argc > 3 && !strcmp(argv[3], "--tolerant");
        std::vector<std::string> cns(edges.size()), errs((size_t)nthreads);
        auto work = [&](int t) {
            try {
                for (size_t e = (size_t)t; e < edges.size(); e += (size_t)nthreads) {
                    const auto& seqs = edges[e];
                    auto alignment_engine = spoa::createAlignmentEngine(static_cast<spoa::AlignmentType>(1), 5, -4, -8);
                    auto graph = spoa::createGraph();
                    for (const std::string& s : seqs) {
                        if (s.empty()) continue;
                        auto alignment = alignment_engine->align_sequence_with_graph(s, graph);
                        graph->add_alignment(alignment, s);
                    }
                    if (!tolerant) cns[e] = graph->generate_consensus();
                    else { try { cns[e] = graph->generate_consensus(); } catch (const std::exception& ex) { cns[e] = std::string("ERROR ") + ex.what(); } }
                }
            } catch (const std::exception& e) { errs[(size_t)t] = e.what(); }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
        work(0);
        for (auto& t : th) t.join();
        for (const std::string& e : errs) if (!e.empty()) throw std::runtime_error(e);
        for (const std::string& c : cns) printf("%s\n", c.c_str());
        const spoa::hx::Stats st = spoa::hx::stats();
        fprintf(stderr, "device_calls=%llu sets=%llu\n", (unsigned long long)st.device_calls, (unsigned long long)st.sets);
        spoa::hx::shutdown();
    } catch (const std::exception& e) {
        fprintf(stderr, "[ERROR] %s\n", e.what());
        return 1;
    }
    return 0;
}
