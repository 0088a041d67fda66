// make_spoa_vectors — produces tests/golden/spoa/*.json from the REAL rvaser/spoa 1.1.3 (the version the reference pins,
// /root/reference/src/haslr_assemble/Makefile:1). NOT built in this repository's image (spoa is not available here, there is no
// network): a maintainer builds it on any networked machine against a genuine libspoa.a and commits the JSON it prints.
//
//   g++ -O2 -std=c++11 -I<spoa>/include make_spoa_vectors.cpp <spoa>/build/lib/libspoa.a -o make_spoa_vectors
//   ./make_spoa_vectors [match mismatch gap] < sequences.txt > tests/golden/spoa/<name>.json
//
// Input: one case per paragraph - ">name", then one ACGT sequence per line (alignment order), a blank line between cases.
// The five calls are exactly the reference's (Assemble.cpp:499,500,539,540,554): kNW, linear gap, sequences added in order, unit weights.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "spoa/spoa.hpp"

static std::string consensus_of(const std::vector<std::string>& seqs, int m, int n, int g) {
    auto engine = spoa::createAlignmentEngine(static_cast<spoa::AlignmentType>(1), (int8_t)m, (int8_t)n, (int8_t)g);
    auto graph = spoa::createGraph();
    size_t used = 0;
    for (const std::string& s : seqs) {
        if (s.empty()) continue;   // Assemble.cpp:537 skips empty sub-sequences
        auto alignment = engine->align_sequence_with_graph(s, graph);
        graph->add_alignment(alignment, s);
        used++;
    }
    return used ? graph->generate_consensus() : std::string();   // Assemble.cpp:544-551
}

int main(int argc, char** argv) {
    const int m = argc > 3 ? atoi(argv[1]) : 5, n = argc > 3 ? atoi(argv[2]) : -4, g = argc > 3 ? atoi(argv[3]) : -8;
    std::vector<std::pair<std::string, std::vector<std::string>>> cases;
    std::string line;
    while (std::getline(std::cin, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') cases.push_back({line.substr(1), {}});
        else if (!cases.empty()) cases.back().second.push_back(line == "-" ? std::string() : line);   // "-" = an empty sequence
    }
    printf("{\"spoa_version\": \"1.1.3\", \"match\": %d, \"mismatch\": %d, \"gap\": %d, \"algorithm\": \"kNW\",\n \"cases\": [", m, n, g);
    for (size_t i = 0; i < cases.size(); i++) {
        printf("%s\n  {\"name\": \"%s\", \"sequences\": [", i ? "," : "", cases[i].first.c_str());
        for (size_t k = 0; k < cases[i].second.size(); k++) printf("%s\"%s\"", k ? ", " : "", cases[i].second[k].c_str());
        printf("], \"consensus\": \"%s\"}", consensus_of(cases[i].second, m, n, g).c_str());
    }
    printf("\n ]}\n");
    return 0;
}
