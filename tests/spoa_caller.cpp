// Test program for include/spoa_hx.hpp: a caller written against the five spoa symbols the reference uses
// (Assemble.cpp:499,500,539,540,554), in the reference's call pattern — one engine + one graph per edge, sequences aligned and added
// one after the other, consensus at the end. Input: edges separated by blank lines, one sequence per line; output: one consensus per line.
// With --batch the same sets go through spoa::hx::consensus_batch in one call.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
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
        for (const auto& seqs : edges) {
            auto alignment_engine = spoa::createAlignmentEngine(static_cast<spoa::AlignmentType>(1), 5, -4, -8);
            auto graph = spoa::createGraph();
            for (const std::string& s : seqs) {
                if (s.empty()) continue;
                auto alignment = alignment_engine->align_sequence_with_graph(s, graph);
                graph->add_alignment(alignment, s);
            }
            std::string consensus = graph->generate_consensus();
            printf("%s\n", consensus.c_str());
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "[ERROR] %s\n", e.what());
        return 1;
    }
    return 0;
}
