// haslr_assemble — drop-in replacement of the reference's CLI for the backbone + consensus stage.
//
// Same command line as the reference binary that bin/haslr.py:66 invokes (Commandline.cpp:77-93):
//   haslr_assemble -t N -c contigs.fa -l reads.fa -m map.paf -d outdir
//                  [--aln-block INT] [--aln-sim FLOAT] [--uniq-dev FLOAT] [--edge-sup INT]
//                  [--long-fofn] [--mapping-fofn] [--version] [-h]
// and the same conventions: exit 0 on success and for -h / --version (haslr.py's check_program relies on it),
// `[ERROR] ...` on stderr + EXIT_FAILURE otherwise, progress on stderr, outputs inside -d.
// Build-only additions: --device INT (HIP device, default 0), --poa-block INT, --gpus INT (the GPUs of this node the per-read and per-edge work
// is spread over: one host thread and one RCCL rank per GPU inside this process, ONE all-gather of the edge-support records; default 1).
// -t is accepted and clamped like the reference's, but the per-read / per-edge work runs on the GPU.
// index.contig / index.longread (the reference's cache files, Contig.cpp:119-159, Longread.cpp:322-372) are written into -d and loaded
// instead of the text inputs when they exist, like main.cpp:39-103 does (host/index_cache.cpp; records read back from index.longread are
// the filtered set and go straight to trim + chain).
#include <unistd.h>
#include <getopt.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/haslr_hip.h"
#include "haslr_host.h"

static const char* kVersion = "0.8a1";   // the reference's prog_version (Commandline.cpp:65): same pipeline stage, same outputs

static double cpu_time() {
    struct rusage t;
    getrusage(RUSAGE_SELF, &t);
    return t.ru_utime.tv_sec + t.ru_utime.tv_usec / 1e6 + t.ru_stime.tv_sec + t.ru_stime.tv_usec / 1e6;
}
static double real_time() {
    struct timeval t;
    gettimeofday(&t, nullptr);
    return t.tv_sec + t.tv_usec / 1e6;
}

// HASLR_STAGE_TIMES=<file>: what a one-shot run spent where, as one JSON object (bench.py's cli_e2e leg reads it; the stderr lines stay the reference's)
struct StageTimes {
    std::vector<std::pair<std::string, double>> v;
    void add(const char* k, double s) { v.emplace_back(k, s); }
    void write(const char* path) const {
        FILE* f = path && *path ? fopen(path, "w") : nullptr;
        if (!f) return;
        fprintf(f, "{");
        for (size_t i = 0; i < v.size(); i++) fprintf(f, "%s\"%s\": %.6f", i ? ", " : "", v[i].first.c_str(), v[i].second);
        fprintf(f, "}\n");
        fclose(f);
    }
};
static uint64_t file_bytes(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode) ? (uint64_t)st.st_size : 0; }

static void help_short() { fprintf(stderr, "usage: haslr_assemble -c contig.fasta -l longread.fasta -m lr2contig.paf -d outdir [options]\n"); }

static void help(const hx_params& p) {
    help_short();
    fprintf(stderr, "\nRequired options:\n");
    fprintf(stderr, "    -c STR            Path to contigs file (also --contig)\n");
    fprintf(stderr, "    -l STR            Path to long read dataset (also --long)\n");
    fprintf(stderr, "    -m STR            Path to mappings of long reads onto contigs (also --mapping)\n");
    fprintf(stderr, "    -d STR            Path to the output directory (also --dir)\n");
    fprintf(stderr, "\nAdvanced options:\n");
    fprintf(stderr, "    --aln-block       Minimum length of alignment block [%d]\n", p.min_aln_block);
    fprintf(stderr, "    --aln-sim         Minimum alignment similarity [%.2lf]\n", p.min_aln_sim);
    fprintf(stderr, "    --uniq-dev        Maximum deviation from mean frequency of uniq contigs [%.2lf]\n", p.max_uniq_dev);
    fprintf(stderr, "    --edge-sup        Minimum number of long read supporting each edge [%d]\n", p.min_edge_sup);
    fprintf(stderr, "\nOther options:\n");
    fprintf(stderr, "    -t INT            Number of CPU cores to use (also --threads)\n");
    fprintf(stderr, "    --long-fofn       The file passed by -l is fofn\n");
    fprintf(stderr, "    --mapping-fofn    The file passed by -m is fofn\n");
    fprintf(stderr, "    --device INT      HIP device to run on [0]\n");
    fprintf(stderr, "    --gpus INT        Number of GPUs of this node to use, devices 0..INT-1 [1]\n");
    fprintf(stderr, "    --version         Prints version (%s)\n", kVersion);
    fprintf(stderr, "    -h                Prints this help message (also --help)\n\n");
}

// tools that write their data from atexit handlers / static destructors (rocprofv3, roctracer, gcov, sanitizers) need the ordinary exit
static bool tooling_attached() {
    for (const char* v : {"HASLR_FULL_TEARDOWN", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB", "ROCTRACER_DOMAIN", "GCOV_PREFIX", "ASAN_OPTIONS", "LSAN_OPTIONS", "TSAN_OPTIONS", "UBSAN_OPTIONS"})
        if (getenv(v)) return true;
    if (const char* pre = getenv("LD_PRELOAD")) if (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "asan")) return true;
    return false;
}

extern char** environ;

int main(int argc, char* argv[]) {
    hx_params prm{500, 0.85, 55, 0.15, 3, 0.0};
    std::string contig_path, long_path, mapping_path, out_dir;
    bool long_fofn = false, mapping_fofn = false;
    unsigned num_threads = 1;
    int device = 0, poa_block = 0, gpus = 1;
    if (argc == 1) { help_short(); return EXIT_FAILURE; }
    static struct option lo[] = {{"contig", required_argument, 0, 'c'}, {"long", required_argument, 0, 'l'}, {"mapping", required_argument, 0, 'm'},
                                 {"dir", required_argument, 0, 'd'}, {"help", no_argument, 0, 'h'}, {"threads", required_argument, 0, 't'},
                                 {"version", no_argument, 0, 0}, {"long-fofn", no_argument, 0, 0}, {"mapping-fofn", no_argument, 0, 0},
                                 {"aln-block", required_argument, 0, 0}, {"aln-sim", required_argument, 0, 0}, {"uniq-dev", required_argument, 0, 0},
                                 {"edge-sup", required_argument, 0, 0}, {"device", required_argument, 0, 0}, {"poa-block", required_argument, 0, 0}, {"gpus", required_argument, 0, 0}, {0, 0, 0, 0}};
    int ch, li;
    while ((ch = getopt_long(argc, argv, "c:l:m:d:t:h", lo, &li)) != -1) {
        switch (ch) {
            case 'c': contig_path = optarg; break;
            case 'l': long_path = optarg; break;
            case 'm': mapping_path = optarg; break;
            case 'd': out_dir = optarg; break;
            case 't': {
                int v = atoi(optarg), hw = (int)std::thread::hardware_concurrency();
                num_threads = v < 1 ? 1 : (v > hw ? (unsigned)hw : (unsigned)v);
                break;
            }
            case 'h': help(prm); return EXIT_SUCCESS;
            case 0:
                if (li == 6) { fprintf(stdout, "%s\n", kVersion); return EXIT_SUCCESS; }
                else if (li == 7) long_fofn = true;
                else if (li == 8) mapping_fofn = true;
                else if (li == 9) { int v = atoi(optarg); prm.min_aln_block = v < 0 ? 500 : (uint32_t)v; }
                else if (li == 10) { prm.min_aln_sim = atof(optarg); if (prm.min_aln_sim < 0 || prm.min_aln_sim > 1) prm.min_aln_sim = 0.85; }
                else if (li == 11) prm.max_uniq_dev = atof(optarg);
                else if (li == 12) { int v = atoi(optarg); prm.min_edge_sup = v < 0 ? 3 : (uint32_t)v; }
                else if (li == 13) device = atoi(optarg);
                else if (li == 14) poa_block = atoi(optarg);
                else if (li == 15) { gpus = atoi(optarg); if (gpus < 1) gpus = 1; }
                else { help_short(); return EXIT_FAILURE; }
                break;
            default: help_short(); return EXIT_FAILURE;
        }
    }
    const char* req[4][2] = {{"c", contig_path.c_str()}, {"l", long_path.c_str()}, {"m", mapping_path.c_str()}, {"d", out_dir.c_str()}};
    for (auto& r : req)
        if (!*r[1]) { fprintf(stderr, "[ERROR] (CommandLine:parseCommandLine) option -%s is required!\n", r[0]); help_short(); return EXIT_FAILURE; }
    errno = 0;
    if (mkdir(out_dir.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH) == -1 && errno != EEXIST) return EXIT_FAILURE;
    if (long_fofn) fprintf(stderr, "[NOTE] file passed by -l is a file of file names (FOFN)\n");
    if (mapping_fofn) fprintf(stderr, "[NOTE] file passed by -m is a file of file names (FOFN)\n");
    fprintf(stderr, "\n[NOTE] number of threads: %d\n\n", num_threads);
    const double c0 = cpu_time(), r0 = real_time();
    auto elapsed = [&]() { fprintf(stderr, "       elapsed time %.2lf CPU seconds (%.2lf real seconds)\n\n", cpu_time() - c0, real_time() - r0); };

    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before HIP initialises: the POA launch classes overlap on separate hardware queues (include/haslr_hip.h)
    StageTimes st;
    // --gpus N > 1 (or HASLR_FORCE_GROUP=1, which sends one GPU through the same code): a group of contexts, one rank per GPU
    const bool grouped = gpus > 1 || getenv("HASLR_FORCE_GROUP");
    hx_ctx* ctx = nullptr;
    hx_group* group = nullptr;
    // The device side of the start-up runs on a thread of its own BESIDE the parse of the text inputs (this program runs once: main.cpp:28-228 of the
    // reference; what a user of haslr.py:66 waits for is the whole run, not the hot path): the HIP runtime and the context(s), the options, and the arena
    // of the consensus workspace (hx_poa_reserve: up to 140 GB for a genome of 10^8 bases and more - seconds of allocation that used to sit inside the
    // consensus stage). The estimate is 64 bytes of workspace per long-read base (12 Mb: 9.2 GB for 3.0e8 bases; 140 Mb: 215 GB for 3.5e9), at most what a call
    // ever settles on; hx_poa_reserve itself stops at 80 % of what is free.
    std::string gpu_error;
    double t_gpu_init = 0, t_reserve = 0;
    const uint64_t lr_bytes = long_fofn ? 0 : file_bytes(long_path);
    // (ranks that may SHARE a device - transport "host", the rehearsal of the multi-GPU logic on a box with fewer GPUs than ranks - reserve nothing: each would
    // take its share of what is free when it looks, and together they would leave no room for the inputs)
    const bool shared_devices = grouped && getenv("HASLR_GROUP_TRANSPORT") && !strcmp(getenv("HASLR_GROUP_TRANSPORT"), "host");
    const uint64_t reserve_bytes = getenv("HASLR_NO_RESERVE") || shared_devices ? 0 : std::min<uint64_t>(232ull << 30, lr_bytes * 64);
    auto gpu_start = [&]() {
        const double g0 = real_time();
        if (grouped) {
            std::vector<int> devs((size_t)gpus);
            for (int r = 0; r < gpus; r++) devs[(size_t)r] = device + r;   // --device with --gpus N: the ranks run on devices device .. device + N - 1
            if (hx_group_create(gpus, device ? devs.data() : nullptr, getenv("HASLR_GROUP_TRANSPORT"), &group) != 0) { gpu_error = hx_last_error(); return; }
            if (const char* ts = getenv("HASLR_GROUP_TIMEOUT_S")) hx_group_set_timeout(group, atof(ts));
            ctx = hx_group_ctx(group, 0);
            if (poa_block) for (int r = 0; r < gpus; r++) hx_set_poa_block(hx_group_ctx(group, r), poa_block);
        } else {
            if (hx_ctx_create(device, nullptr, &ctx) != 0) { gpu_error = hx_last_error(); return; }
            if (poa_block) hx_set_poa_block(ctx, poa_block);
        }
        // The library reads no environment: this APPLICATION hands the HX_* variables that name library options (hx_option_names) to its contexts, once
        const std::string names = std::string(",") + hx_option_names() + ",prof1,prof2,prof3,";
        for (char** ev = environ; ev && *ev; ev++) {
            if (strncmp(*ev, "HX_", 3) != 0) continue;
            const char* eq = strchr(*ev, '=');
            if (!eq) continue;
            std::string key((const char*)*ev + 3, eq);
            for (char& ch : key) ch = (char)tolower((unsigned char)ch);
            if (names.find("," + key + ",") == std::string::npos) { fprintf(stderr, "[WARNING] %.*s is not an option of this build (ignored)\n", (int)(eq - *ev), *ev); continue; }
            for (int r = 0; r < (grouped ? gpus : 1); r++)
                if (hx_set_option(grouped ? hx_group_ctx(group, r) : ctx, key.c_str(), eq + 1) != 0) { gpu_error = hx_last_error(); return; }
        }
        t_gpu_init = real_time() - g0;
        if (reserve_bytes) {
            const double r0 = real_time();
            std::vector<std::thread> rs;
            for (int r = 0; r < (grouped ? gpus : 1); r++) rs.emplace_back([&, r]() { (void)hx_poa_reserve(grouped ? hx_group_ctx(group, r) : ctx, reserve_bytes); });   // (best effort: the consensus call allocates what is missing)
            for (auto& t : rs) t.join();
            t_reserve = real_time() - r0;
        }
    };
    std::thread gpu_thread(gpu_start);
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } gpu_joiner{gpu_thread};

    fprintf(stderr, "[NOTE] loading contig sequences, long read sequences and alignments...\n");
    // index.contig / index.longread of the output directory are loaded when they exist, written when they do not (main.cpp:39-103)
    int used_ci = 0, used_li = 0;
    const double tl0 = real_time();
    hxh_dataset* ds = hxh_dataset_load_cached(out_dir.c_str(), contig_path.c_str(), long_path.c_str(), long_fofn, mapping_path.c_str(), mapping_fofn, num_threads /* -t */,
                                              &used_ci, &used_li);
    if (!ds) { fprintf(stderr, "%s\n", hxh_last_error()); return EXIT_FAILURE; }
    st.add("load_s", real_time() - tl0);
    if (used_ci) fprintf(stderr, "[NOTE] reading contig index: %s/index.contig...\n", out_dir.c_str());
    if (used_li) fprintf(stderr, "[NOTE] reading long read and alignment index: %s/index.longread...\n", out_dir.c_str());
    {
        const double t0 = real_time();
        if (!used_ci && hxh_dataset_write_contig_index(ds, (out_dir + "/index.contig").c_str()) != 0) { fprintf(stderr, "%s\n", hxh_last_error()); return EXIT_FAILURE; }
        st.add("index_contig_write_s", real_time() - t0);
    }
    hx_contigs vc; hx_reads vr; hx_hits vh; const uint64_t* rho;
    hxh_dataset_views(ds, &vc, &vr, &vh, &rho);
    // the reference prints ONE "loaded N alignments" line, N = the records that survive the load-time filters (main.cpp:70,103). From an
    // index.longread that is the record count; from a PAF the filters run on the GPU in the next stage, which prints the line then.
    fprintf(stderr, "       loaded %u contigs\n       loaded %u long reads\n", vc.n, vr.n);
    if (used_li) fprintf(stderr, "       loaded %lu alignments\n", (unsigned long)vh.n);
    prm.uniq_freq = hxh_dataset_uniq_freq(ds);
    fprintf(stderr, "[NOTE] calculating kmer frequency of unique contigs\n       mean: %.2lf\n", prm.uniq_freq);
    {   // the device side of the start-up has had the whole parse to itself
        const double t0 = real_time();
        gpu_thread.join();
        st.add("gpu_init_s", t_gpu_init); st.add("workspace_reserve_s", t_reserve); st.add("gpu_start_wait_after_load_s", real_time() - t0);
        if (!gpu_error.empty()) { fprintf(stderr, "[ERROR] %s\n", gpu_error.c_str()); return EXIT_FAILURE; }
        if (grouped) fprintf(stderr, "[NOTE] %d GPU ranks in this process, edge-record exchange over %s\n", gpus, hx_group_transport(group));
    }
    elapsed();
    if (grouped) {
        // inputs are replicated on every GPU (they fit: DESIGN.md 3); reads are sharded by id range, edges by estimated DP cost
        std::vector<uint32_t> bounds((size_t)gpus + 1);
        hxh_shard_bounds(ds, (uint32_t)gpus, bounds.data());
        std::vector<std::string> uerr((size_t)gpus);
        std::vector<std::thread> up;
        for (int r = 0; r < gpus; r++)
            up.emplace_back([&, r]() {
                hx_ctx* c = hx_group_ctx(group, r);
                if (hx_upload(c, &vc, &vr, &vh, rho) != 0 || hx_set_read_shard(c, bounds[r], bounds[r + 1]) != 0) uerr[r] = hx_last_error();
                hx_set_prefiltered(c, used_li);
            });
        for (auto& t : up) t.join();
        for (int r = 0; r < gpus; r++) if (!uerr[r].empty()) { fprintf(stderr, "[ERROR] rank %d: %s\n", r, uerr[r].c_str()); return EXIT_FAILURE; }
        std::vector<hx_backend> tables((size_t)gpus);
        std::vector<hxh_run*> runs((size_t)gpus);
        for (int r = 0; r < gpus; r++) {
            hx_group_backend_fill(group, r, &tables[r]);
            runs[r] = hxh_run_create(ds, &prm, &tables[r], r == 0 ? out_dir.c_str() : nullptr);
        }
        static const char* notes[5] = {"[NOTE] fixing overlapping alignments and building compact long reads...", "[NOTE] building and cleaning the backbone graph...",
                                       "[NOTE] calculating long read coordinates between anchors...", "[NOTE] calling consensus sequence between anchors...",
                                       "[NOTE] generating the assembly from the cleaned backbone graph..."};
        struct Cb { decltype(elapsed)* el; hx_group* g; std::vector<hxh_run*>* runs; bool from_paf; } cb{&elapsed, group, &runs, !used_li};
        auto on_stage = [](int s, int begin, void* u) {
            Cb* c = (Cb*)u;
            if (begin) fprintf(stderr, "%s\n", notes[s]);
            else {
                if (s == 0 && c->from_paf) {   // (the count the reference prints at load time, main.cpp:70,103: the ranks' filtered sets together)
                    unsigned long n = 0;
                    for (hxh_run* r : *c->runs) n += (unsigned long)hxh_run_chain_out(r)->n_aln;
                    fprintf(stderr, "       loaded %lu alignments\n", n);
                }
                if (s == 1) { uint64_t by = 0; double ms = 0; hx_group_exchange_stats(c->g, &by, &ms); fprintf(stderr, "       exchanged %lu bytes of edge records in %.2f ms (%s)\n", (unsigned long)by, ms, hx_group_transport(c->g)); }
                (*c->el)();
            }
        };
        if (hxh_runs_all_sharded(runs.data(), (uint32_t)gpus, bounds.data(), on_stage, &cb) != 0) { fprintf(stderr, "[ERROR] %s\n", hxh_last_error()); return EXIT_FAILURE; }
        // index.longread: the ranks' filtered alignments in rank order (written after the stages here: every rank's chain output is still held)
        if (!used_li && hxh_runs_write_longread_index(runs.data(), (uint32_t)gpus, (out_dir + "/index.longread").c_str()) != 0) { fprintf(stderr, "%s\n", hxh_last_error()); return EXIT_FAILURE; }
        fprintf(stderr, "[NOTE] cleaning up the memory!\n");
        fprintf(stderr, "[NOTE] elapsed time %.2lf CPU seconds (%.2lf real seconds)\n\n*** BYE ***\n\n", cpu_time() - c0, real_time() - r0);
        fflush(nullptr);
        if (tooling_attached()) { for (hxh_run* r : runs) hxh_run_free(r); hxh_dataset_free(ds); hx_group_destroy(group); return EXIT_SUCCESS; }   // (profilers, sanitizers, leak checks: the ordinary exit)
        _exit(EXIT_SUCCESS);
    }
    {
        const double t0 = real_time();
        if (hx_upload(ctx, &vc, &vr, &vh, rho) != 0) { fprintf(stderr, "[ERROR] %s\n", hx_last_error()); return EXIT_FAILURE; }
        st.add("upload_s", real_time() - t0);
    }
    hx_set_prefiltered(ctx, used_li);

    hx_backend be;
    hx_backend_fill(ctx, &be);
    hxh_run* run = hxh_run_create(ds, &prm, &be, out_dir.c_str());
    hxh_run_set_async_writers(run, 1);   // the GFA snapshots are written beside the GPU stages; the last stage waits for them
    struct { const char* note; int (*fn)(hxh_run*); } stages[] = {
        {"[NOTE] fixing overlapping alignments and building compact long reads...", hxh_run_chain},
        {"[NOTE] building and cleaning the backbone graph...", hxh_run_graph},
        {"[NOTE] calculating long read coordinates between anchors...", hxh_run_coords},
        {"[NOTE] calling consensus sequence between anchors...", hxh_run_consensus},
        {"[NOTE] generating the assembly from the cleaned backbone graph...", hxh_run_assemble}};
    // index.longread: written when the filtered set is known (the reference writes it at load time, main.cpp:65-89) - on a thread of its own beside the
    // later stages (round 6: the default; it is a gigabyte at 140 Mb and nothing later reads it), joined before the program ends. HASLR_INDEX_SYNC=1
    // writes it before the GPU stages go on, as before.
    std::thread index_writer;
    std::string index_error;
    auto finish_index = [&]() -> bool {
        if (index_writer.joinable()) index_writer.join();
        if (!index_error.empty()) { fprintf(stderr, "%s\n", index_error.c_str()); return false; }
        return true;
    };
    static const char* stage_keys[5] = {"chain_s", "graph_s", "coords_s", "consensus_s", "assemble_s"};
    int stage_no = 0;
    for (auto& sg : stages) {
        fprintf(stderr, "%s\n", sg.note);
        const double ts0 = real_time();
        if (sg.fn(run) != 0) {
            fprintf(stderr, "[ERROR] %s\n", hxh_last_error());
            finish_index();
            hxh_run_free(run);   // joins the GFA writer threads: nothing streams into backbone.0x.gfa while the static destructors run
            return EXIT_FAILURE;
        }
        st.add(stage_keys[stage_no++], real_time() - ts0);
        if (sg.fn == hxh_run_chain && !used_li) fprintf(stderr, "       loaded %lu alignments\n", (unsigned long)hxh_run_chain_out(run)->n_aln);   // (the count the reference prints at load time)
        if (sg.fn == hxh_run_chain && !used_li) {
            const std::string path = out_dir + "/index.longread";
            const double tw0 = real_time();
            if (!getenv("HASLR_INDEX_SYNC"))
                index_writer = std::thread([run, path, &index_error]() {
                    if (hxh_run_write_longread_index(run, path.c_str()) != 0) { index_error = hxh_last_error(); if (index_error.empty()) index_error = "[ERROR] could not write " + path; remove(path.c_str()); }
                });
            else if (hxh_run_write_longread_index(run, path.c_str()) != 0) { fprintf(stderr, "%s\n", hxh_last_error()); return EXIT_FAILURE; }
            st.add("index_longread_write_sync_s", real_time() - tw0);
        }
        elapsed();
    }
    {
        const double t0 = real_time();
        if (!finish_index()) return EXIT_FAILURE;
        st.add("index_longread_join_s", real_time() - t0);
    }
    st.add("total_s", real_time() - r0);
    st.write(getenv("HASLR_STAGE_TIMES"));
    fprintf(stderr, "[NOTE] cleaning up the memory!\n");
    if (!tooling_attached()) {   // everything is written and closed: the release of up to ~250 GB of device memory and of the host arrays is left
                                            // to the end of the process (1.5-2 s of a 10 s run at 140 Mb); HASLR_FULL_TEARDOWN=1 frees object by object (leak checks)
        fprintf(stderr, "[NOTE] elapsed time %.2lf CPU seconds (%.2lf real seconds)\n\n*** BYE ***\n\n", cpu_time() - c0, real_time() - r0);
        fflush(nullptr);
        _exit(EXIT_SUCCESS);
    }
    hxh_run_free(run);
    hxh_dataset_free(ds);
    hx_ctx_destroy(ctx);
    fprintf(stderr, "[NOTE] elapsed time %.2lf CPU seconds (%.2lf real seconds)\n\n*** BYE ***\n\n", cpu_time() - c0, real_time() - r0);
    return EXIT_SUCCESS;
}
