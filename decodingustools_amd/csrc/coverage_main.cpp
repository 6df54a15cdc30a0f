// dut-coverage -- the `coverage`, `find-y-branch` and `find-mt-branch` subcommands of the reference CLI;
// `coverage` is the default: (src/cli.rs:14-61, src/main.rs:36-70)
// on the MI355X engine.  Same flags and defaults; BED to -o, the CoverageOutput JSON to ./summary.json.
// -s/--summary: the HTML report (the reference's sections and numbers in this project's own markup); the
// per-contig coverage figures <contig>_coverage.svg go beside the BED file.
#include "../../include/dut_bam.h"
#include "../../include/dut_haplogroup.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <string>
#include <cerrno>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

static void usage()
{
    fprintf(stderr,
            "Usage: dut-coverage [coverage] <BAM_FILE> -r <REFERENCE_FILE> [-o callable_regions.bed] [-s summary.html]\n"
            "       [-L <CONTIG>]... [--min-depth 4] [--max-depth 500] [--min-mapping-quality 10]\n"
            "       [--min-base-quality 20] [--min-depth-for-low-mapq 10] [--max-low-mapq 1]\n"
            "       [--max-low-mapq-fraction 0.1] [--device 0 | --devices 0,1,...]\n");
}

// find-y-branch / find-mt-branch (src/cli.rs:62-105, src/commands/find_branch.rs).  The reference
// downloads the tree; here --tree names a local JSON file of the provider's shape.
static int find_branch_main(int argc, char **argv, int tree_type)
{
    std::string bam, ref, out, tree;
    uint32_t min_depth = 10; unsigned min_quality = 20;
    int provider = DUT_PROVIDER_FTDNA, show_snps = 0, device = 0;
    auto usage_fb = [&]() {
        fprintf(stderr, "Usage: dut-coverage %s <BAM_FILE> -r <REFERENCE_FILE> <OUTPUT_FILE> --tree <TREE_JSON>\n"
                        "       [--min-depth 10] [--min-quality 20] [--provider ftdna|decodingus] [--show-snps] [--device 0]\n",
                tree_type == DUT_TREE_YDNA ? "find-y-branch" : "find-mt-branch");
    };
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i], val;
        const size_t eq = a.find('=');
        const bool has_eq = a.rfind("--", 0) == 0 && eq != std::string::npos;
        if (has_eq) { val = a.substr(eq + 1); a = a.substr(0, eq); }
        auto next = [&]() -> const char * {
            if (has_eq) return val.c_str();
            if (i + 1 >= argc) { usage_fb(); exit(2); }
            return argv[++i];
        };
        if (a == "-r" || a == "--reference") ref = next();
        else if (a == "--tree") tree = next();
        else if (a == "--min-depth") min_depth = (uint32_t)strtoul(next(), nullptr, 10);
        else if (a == "--min-quality") min_quality = (unsigned)strtoul(next(), nullptr, 10);
        else if (a == "--provider") {
            const std::string p = next();
            if (p == "ftdna") provider = DUT_PROVIDER_FTDNA;
            else if (p == "decodingus") provider = DUT_PROVIDER_DECODINGUS;
            else { fprintf(stderr, "error: invalid value '%s' for '--provider'\n", p.c_str()); return 2; }
        }
        else if (a == "--show-snps") show_snps = 1;
        else if (a == "--device") device = atoi(next());
        else if (a == "-h" || a == "--help") { usage_fb(); return 0; }
        else if (!a.empty() && a[0] != '-' && bam.empty()) bam = a;
        else if (!a.empty() && a[0] != '-' && out.empty()) out = a;
        else { fprintf(stderr, "error: unexpected argument '%s'\n", argv[i]); usage_fb(); return 2; }
    }
    if (bam.empty() || ref.empty() || out.empty() || tree.empty()) { usage_fb(); return 2; }
    char err[1024] = {0};
    const int rc = dut_find_branch_files(bam.c_str(), ref.c_str(), tree.c_str(), out.c_str(), min_depth, (uint8_t)min_quality,
                                         tree_type, provider, show_snps, device, err, sizeof(err));
    if (rc != CL_OK) { fprintf(stderr, "Error: %s\n", err); return 1; }
    fflush(nullptr);
    _exit(0);                              // outputs are closed; skip the HIP runtime's exit handlers (see main)
}

// DUT_TIMING=1: the wall clock (CLOCK_REALTIME, seconds) at the start of main and right before the process leaves, so
// that a harness that started the tool can tell what the loader took before main and what the exit took after it
static void stamp(const char *what)
{
    const char *e = getenv("DUT_TIMING");
    if (!e || *e != '1') return;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    fprintf(stderr, "[dut-timing] wall clock at %s: %.6f\n", what, (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
}

int main(int argc, char **argv)
{
    stamp("main");
    if (argc > 1 && !strcmp(argv[1], "find-y-branch")) return find_branch_main(argc, argv, DUT_TREE_YDNA);
    if (argc > 1 && !strcmp(argv[1], "find-mt-branch")) return find_branch_main(argc, argv, DUT_TREE_MTDNA);
    cl_options opt = {4, 500, 10, 20, 10, 1, 0.1};      // src/cli.rs:34-60
    std::string bam, ref, out = "callable_regions.bed", summary = "summary.html";
    std::vector<const char *> contigs;
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        std::string val;
        const size_t eq = a.find('=');
        const bool has_eq = a.rfind("--", 0) == 0 && eq != std::string::npos;
        if (has_eq) { val = a.substr(eq + 1); a = a.substr(0, eq); }
        auto next = [&]() -> const char * {
            if (has_eq) return val.c_str();
            if (i + 1 >= argc) { usage(); exit(2); }
            return argv[++i];
        };
        if (a == "coverage" && bam.empty()) continue;
        else if (a == "-r" || a == "--reference") ref = next();
        else if (a == "-o" || a == "--output") out = next();
        else if (a == "-s" || a == "--summary") summary = next();
        else if (a == "-L" || a == "--contig") contigs.push_back(strdup(next()));
        else if (a == "--min-depth") opt.min_depth = (uint32_t)strtoul(next(), nullptr, 10);
        else if (a == "--max-depth") opt.max_depth = (uint32_t)strtoul(next(), nullptr, 10);
        else if (a == "--min-mapping-quality") opt.min_mapping_quality = (uint8_t)strtoul(next(), nullptr, 10);
        else if (a == "--min-base-quality") opt.min_base_quality = (uint8_t)strtoul(next(), nullptr, 10);
        else if (a == "--min-depth-for-low-mapq") opt.min_depth_for_low_mapq = (uint32_t)strtoul(next(), nullptr, 10);
        else if (a == "--max-low-mapq") opt.max_low_mapq = (uint8_t)strtoul(next(), nullptr, 10);
        else if (a == "--max-low-mapq-fraction") opt.max_low_mapq_fraction = strtod(next(), nullptr);
        else if (a == "--device") devices.assign(1, atoi(next()));
        else if (a == "--devices") {
            // the contigs are dealt to these devices (HIP ordinals, comma separated; an ordinal may repeat)
            devices.clear();
            const std::string list = next();
            for (size_t b = 0; b <= list.size();) {
                const size_t e = std::min(list.find(',', b), list.size());
                if (e > b) devices.push_back(atoi(list.substr(b, e - b).c_str()));
                b = e + 1;
            }
            if (devices.empty()) { fprintf(stderr, "error: invalid value '%s' for '--devices'\n", list.c_str()); return 2; }
        }
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (!a.empty() && a[0] != '-' && bam.empty()) bam = a;
        else { fprintf(stderr, "error: unexpected argument '%s'\n", argv[i]); usage(); return 2; }
    }
    if (bam.empty() || ref.empty()) { usage(); return 2; }
    if (devices.empty()) devices.push_back(0);
    // The analysis runs in a child process and this one returns as soon as the child reports that every output file is
    // written and closed: what is left then -- the kernel taking a few gigabytes of decode buffers, the pinned staging
    // memory and the device context apart, 0.1 to 0.5 s depending on the box -- happens in the background, after the
    // command has returned (the child closes its output streams first, so a caller that reads them to their end does
    // not wait for it either).  Forked before anything touches the GPU.  DUT_CLI_FOREGROUND=1: one process, the caller
    // waits for the teardown too.
    int status_fd = -1;
    {
        const char *fg = getenv("DUT_CLI_FOREGROUND");
        int fds[2];
        if (!(fg && *fg == '1') && pipe(fds) == 0) {
            const pid_t pid = fork();
            if (pid > 0) {
                close(fds[1]);
                unsigned char code = 0;
                ssize_t g;
                do { g = read(fds[0], &code, 1); } while (g < 0 && errno == EINTR);
                if (g == 1) { stamp("return (the child goes on releasing)"); _exit(code); }
                int st = 0;                                   // the child ended without a word: its own status tells
                while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
                _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);
            }
            if (pid == 0) { close(fds[0]); status_fd = fds[1]; }
            else { close(fds[0]); close(fds[1]); }            // no fork: in the foreground
        }
    }
    auto leave = [&](int code) {
        fflush(nullptr);
        if (status_fd >= 0) {
            const unsigned char b = (unsigned char)code;
            if (write(status_fd, &b, 1) != 1) {}
            close(status_fd);
            close(1); close(2);                               // readers of the tool's output see its end now
        }
        _exit(code);
    };
    char err[1024] = {0};
    // this process ends with the analysis: what the library holds (device contexts, readers, decode buffers) is left to
    // the exit (DUT_CLI_TEARDOWN=1: given back piece by piece first, as a library caller's process would)
    const char *td = getenv("DUT_CLI_TEARDOWN");
    const unsigned flags = (td && *td == '1') ? 0u : DUT_FILES_LEAVE_TO_EXIT;
    const int rc = dut_coverage_files_multi(bam.c_str(), ref.c_str(), out.c_str(), "summary.json", summary.c_str(), &opt,
                                            contigs.empty() ? nullptr : contigs.data(), contigs.size(), devices.data(), devices.size(),
                                            flags, err, sizeof(err));
    if (rc != CL_OK) { fprintf(stderr, "Error: Analysis error: %s\n", err); leave(1); }
    // every output file is written and closed: leave without the HIP runtime's exit handlers
    stamp("exit");
    leave(0);
    return 0;
}
