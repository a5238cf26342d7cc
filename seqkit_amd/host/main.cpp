// main.cpp -- `seqkit-hip seq|fx2tab|stats|subseq|convert [flags] [files...]`: command line of the
// reference commands on the hot path (flag names and letters as registered in
// seqkit/cmd/root.go:65-72, seq.go:291-306, fx2tab.go:157-164, stat.go:292-294, subseq.go:549-564,
// convert.go:280-291).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host.h"

using namespace host;

namespace {

struct Flag { const char* s; const char* l; bool takes_value; };

// cobra/pflag syntax: -x, -xyz (bundled booleans), -x VALUE, -xVALUE, --long, --long VALUE, --long=VALUE
struct Parser {
    std::vector<Flag> flags;
    std::vector<std::pair<std::string, std::string>> seen;   // long name -> value ("" for booleans)
    std::vector<std::string> args;
    std::string error;

    const Flag* by_short(char c) const { for (auto& f : flags) if (f.s && f.s[0] == c) return &f; return nullptr; }
    const Flag* by_long(const std::string& n) const { for (auto& f : flags) if (n == f.l) return &f; return nullptr; }

    bool parse(int argc, char** argv, int first)
    {
        for (int i = first; i < argc; i++) {
            std::string a = argv[i];
            if (a == "--") { for (i++; i < argc; i++) args.push_back(argv[i]); break; }
            if (a.size() < 2 || a[0] != '-') { args.push_back(a); continue; }
            if (a[1] == '-') {
                std::string name = a.substr(2), val;
                const size_t eq = name.find('=');
                const bool has_eq = eq != std::string::npos;
                if (has_eq) { val = name.substr(eq + 1); name = name.substr(0, eq); }
                const Flag* f = by_long(name);
                if (!f) { error = "unknown flag: --" + name; return false; }
                if (f->takes_value && !has_eq) {
                    if (i + 1 >= argc) { error = "flag needs an argument: --" + name; return false; }
                    val = argv[++i];
                }
                seen.emplace_back(f->l, f->takes_value ? val : (has_eq ? val : "true"));
                continue;
            }
            for (size_t k = 1; k < a.size(); k++) {
                const Flag* f = by_short(a[k]);
                if (!f) { error = std::string("unknown shorthand flag: '") + a[k] + "' in " + a; return false; }
                if (!f->takes_value) { seen.emplace_back(f->l, "true"); continue; }
                std::string val = a.substr(k + 1);
                if (!val.empty() && val[0] == '=') val = val.substr(1);
                if (val.empty()) {
                    if (i + 1 >= argc) { error = std::string("flag needs an argument: '") + a[k] + "' in " + a; return false; }
                    val = argv[++i];
                }
                seen.emplace_back(f->l, val);
                break;
            }
        }
        return true;
    }
    bool has(const char* l) const { for (auto& kv : seen) if (kv.first == l && kv.second != "false") return true; return false; }
    std::string value(const char* l, const std::string& dflt) const
    {
        std::string v = dflt;
        for (auto& kv : seen) if (kv.first == l) v = kv.second;
        return v;
    }
    std::vector<std::string> values(const char* l) const     // StringSlice: repeated, comma-separated
    {
        std::vector<std::string> out;
        for (auto& kv : seen) {
            if (kv.first != l) continue;
            size_t a = 0;
            for (;;) {
                const size_t c = kv.second.find(',', a);
                out.push_back(kv.second.substr(a, c == std::string::npos ? std::string::npos : c - a));
                if (c == std::string::npos) break;
                a = c + 1;
            }
        }
        return out;
    }
};

int usage()
{
    fprintf(stderr,
            "seqkit-hip -- SeqKit's seq / fx2tab / stats with the per-record work on an AMD MI355X\n\n"
            "Usage:\n  seqkit-hip seq    [-r] [-p] [-g] [-G letters] [--dna2rna] [--rna2dna] [-l] [-u] [-n] [-s] [-q] [-i]\n"
            "                    [-m N] [-M N] [-v] [-V N] [global flags] [files...]\n"
            "  seqkit-hip fx2tab [-l] [-g] [-G] [-B bases]... [-n] [-i] [-H] [-a] [global flags] [files...]\n"
            "  seqkit-hip stats  [-a] [-T] [-G gap-letters] [global flags] [files...]\n"
            "  seqkit-hip subseq [-r start:end | --gtf FILE | --bed FILE] [--chr IDs] [--feature types] [--gtf-tag TAG]\n"
            "                    [-u N] [-d N] [-f] [global flags] [files...]\n"
            "  seqkit-hip convert [--from ENC] [--to ENC] [-f] [-d] [-n N] [-N N] [-F frac] [global flags] [files...]\n\n"
            "Global flags: -t/--seq-type dna|rna|protein|unlimit|auto, -j/--threads N, -w/--line-width N,\n"
            "  -o/--out-file FILE, --quiet, --alphabet-guess-seq-length N, --device N, --gpus N | --devices a,b,..\n"
            "  (text blocks dealt to several GPUs; stats totals merged by an RCCL all-reduce),\n"
            "  --batch-bytes N (text block / batch size), --host-parse (scan and print records on the host)\n");
    return 255;
}

}  // namespace

namespace {
// SKH_TIMING=1: how long main() ran, on stderr; what the process took beyond that is the dynamic loader
// before main and the HIP runtime's own tear-down after it
struct MainTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~MainTimer()
    {
        if (getenv("SKH_TIMING"))
            fprintf(stderr, "[timing] main %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};
}  // namespace

int main(int argc, char** argv)
{
    MainTimer main_timer;
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    const bool is_seq = cmd == "seq", is_fx = cmd == "fx2tab", is_st = cmd == "stats" || cmd == "stat",
               is_sub = cmd == "subseq", is_cv = cmd == "convert";
    if (!is_seq && !is_fx && !is_st && !is_sub && !is_cv) return usage();

    Parser p;
    p.flags = {{"t", "seq-type", true}, {"j", "threads", true}, {"w", "line-width", true}, {"o", "out-file", true},
               {nullptr, "quiet", false}, {nullptr, "alphabet-guess-seq-length", true}, {nullptr, "device", true},
               {nullptr, "gpus", true}, {nullptr, "devices", true},
               {nullptr, "id-regexp", true}, {nullptr, "id-ncbi", false},
               {nullptr, "batch-bytes", true}, {nullptr, "batch-records", true}, {nullptr, "host-parse", false}};
    if (is_seq) {
        const Flag f[] = {{"r", "reverse", false}, {"p", "complement", false}, {"n", "name", false}, {"s", "seq", false},
                          {"q", "qual", false}, {"i", "only-id", false}, {"m", "min-len", true}, {"M", "max-len", true},
                          {"v", "validate-seq", false}, {"V", "validate-seq-length", true},
                          {"g", "remove-gaps", false}, {"G", "gap-letters", true}, {"l", "lower-case", false},
                          {"u", "upper-case", false}, {nullptr, "dna2rna", false}, {nullptr, "rna2dna", false}};
        p.flags.insert(p.flags.end(), f, f + 16);
    } else if (is_cv) {
        const Flag f[] = {{nullptr, "from", true}, {nullptr, "to", true}, {"d", "dry-run", false}, {"f", "force", false},
                          {"n", "nrecords", true}, {"N", "thresh-B-in-n-most-common", true},
                          {"F", "thresh-illumina1.5-frac", true}};
        p.flags.insert(p.flags.end(), f, f + 7);
    } else if (is_sub) {
        const Flag f[] = {{"r", "region", true}, {nullptr, "gtf", true}, {nullptr, "bed", true}, {nullptr, "gtf-tag", true},
                          {nullptr, "chr", true}, {nullptr, "feature", true}, {"u", "up-stream", true},
                          {"d", "down-stream", true}, {"f", "only-flank", false}};
        p.flags.insert(p.flags.end(), f, f + 9);
    } else if (is_fx) {
        const Flag f[] = {{"l", "length", false}, {"g", "gc", false}, {"G", "gc-skew", false}, {"B", "base-content", true},
                          {"i", "only-id", false}, {"n", "name", false}, {"H", "header-line", false}, {"a", "alphabet", false}};
        p.flags.insert(p.flags.end(), f, f + 8);
    } else {
        const Flag f[] = {{"T", "tabular", false}, {"G", "gap-letters", true}, {"a", "all", false}};
        p.flags.insert(p.flags.end(), f, f + 3);
    }
    if (!p.parse(argc, argv, 2)) { fprintf(stderr, "Error: %s\n", p.error.c_str()); return 255; }

    GlobalOpts g;
    std::string t = p.value("seq-type", "auto");
    for (auto& ch : t) ch = (char)tolower((unsigned char)ch);
    g.alphabet = t == "dna" ? DNAredundant : t == "rna" ? RNAredundant : t == "protein" ? Protein
               : t == "unlimit" ? Unlimit : Auto;                         // helper.go:138-156
    g.threads = atoi(p.value("threads", "2").c_str());
    if (g.threads <= 0) return fatal("value of flag --threads should be greater than 0");
    g.line_width = atoi(p.value("line-width", "60").c_str());
    if (g.line_width < 0) return fatal("value of flag --line-width should be greater than 0");
    g.out_file = p.value("out-file", "-");
    g.quiet = p.has("quiet");
    const long long gl = atoll(p.value("alphabet-guess-seq-length", "10000").c_str());
    if (gl < 0) return fatal("value of flag --alphabet-guess-seq-length should be greater than 0");
    if (gl > 0 && gl < 1000) return fatal("value of flag --alphabet-guess-seq-length too small, should >= 1000");
    g.guess_len = (size_t)gl;
    g.device = atoi(p.value("device", "0").c_str());
    // --gpus N: the file's text blocks are dealt to N GPUs in turn (records are independent; `stats`
    // merges the per-GPU totals with one RCCL all-reduce); --devices lists the ordinals instead
    if (p.has("devices")) {
        for (const std::string& d : p.values("devices")) g.devices.push_back(atoi(d.c_str()));
    } else {
        const int n = atoi(p.value("gpus", "1").c_str());
        if (n < 1 || n > 64) return fatal("value of flag --gpus should be between 1 and 64");
        for (int k = 0; k < n; k++) g.devices.push_back(g.device + k);
    }
    if (g.devices.empty()) return fatal("value of flag --devices should not be empty");
    g.device = g.devices[0];
    g.batch_bytes = strtoull(p.value("batch-bytes", "268435456").c_str(), nullptr, 10);
    g.batch_records = strtoull(p.value("batch-records", "4194304").c_str(), nullptr, 10);
    g.host_parse = p.has("host-parse");
    if (p.has("batch-bytes")) g.text_block_bytes = g.batch_bytes;
    if (p.has("id-ncbi")) g.id_regexp = "\\|([^\\|]+)\\| ";                  // helper.go:129-131
    else if (p.has("id-regexp")) g.id_regexp = p.value("id-regexp", "");
    if (g.id_regexp == "^([^\\s]+)\\s?") g.id_regexp.clear();                // fastx.DefaultIDRegexp
    if (!g.id_regexp.empty()) {
        if (g.id_regexp.find('(') == std::string::npos || g.id_regexp.find(')') == std::string::npos)
            return fatal("regular expression must contain \"(\" and \")\" to capture matched ID. default: `^([^\\s]+)\\s?`");
        if (g.host_parse) return fatal("--id-regexp / --id-ncbi are not available with --host-parse");
        // The reference compiles the expression with Go's regexp (RE2 syntax); here it runs on std::regex
        // (ECMAScript).  What the two spell differently is refused rather than matched differently:
        // inline flags and named groups "(?i)", "(?P<name>", and the escapes \p \P \z \Q \C.
        for (size_t i = 0; i + 1 < g.id_regexp.size(); i++) {
            const char a = g.id_regexp[i], b = g.id_regexp[i + 1];
            const bool flags = a == '(' && b == '?' && i + 2 < g.id_regexp.size() && g.id_regexp[i + 2] != ':' &&
                               g.id_regexp[i + 2] != '=' && g.id_regexp[i + 2] != '!' && (i == 0 || g.id_regexp[i - 1] != '\\');
            const bool esc = a == '\\' && (b == 'p' || b == 'P' || b == 'z' || b == 'Q' || b == 'C');
            if (esc) { i++; }
            if (flags || esc)
                return fatal("error parsing regexp: RE2-only syntax (inline flags, named groups, \\p \\P \\z \\Q \\C) is not supported "
                             "by --id-regexp here");
            if (a == '\\') i++;                      // an escaped character is not the start of anything
        }
        try { g.id_re = std::make_shared<std::regex>(g.id_regexp, std::regex::ECMAScript | std::regex::optimize); }
        catch (const std::regex_error& e) { return fatal(std::string("error parsing regexp: ") + e.what()); }
    }
    std::vector<std::string> files = p.args;
    if (files.empty()) files.push_back("-");                                // helper.go:55-56

    if (is_seq) {
        SeqOpts o;
        o.reverse = p.has("reverse"); o.complement = p.has("complement"); o.only_name = p.has("name");
        o.only_seq = p.has("seq"); o.only_qual = p.has("qual"); o.only_id = p.has("only-id");
        o.min_len = atoi(p.value("min-len", "-1").c_str());
        o.max_len = atoi(p.value("max-len", "-1").c_str());
        if (o.min_len >= 0 && o.max_len >= 0 && o.min_len > o.max_len)
            return fatal("value of flag -m (--min-len) should be >= value of flag -M (--max-len)");
        o.remove_gaps = p.has("remove-gaps"); o.gap_letters = p.value("gap-letters", "- .");
        o.lower_case = p.has("lower-case"); o.upper_case = p.has("upper-case");
        o.dna2rna = p.has("dna2rna"); o.rna2dna = p.has("rna2dna");
        if (o.gap_letters.empty()) return fatal("value of flag -G (--gap-letters) should not be empty");         // seq.go:71-78
        for (unsigned char c : o.gap_letters)
            if (c > 127) return fatal("value of -G (--gap-letters) contains non-ASCII characters");
        o.validate = p.has("validate-seq");
        const long long vl = atoll(p.value("validate-seq-length", "10000").c_str());
        if (vl < 0) return fatal("value of flag --validate-seq-length should be greater than 0");
        if (vl > 0 && vl < 1000) return fatal("value of flag --validate-seq-length too small, should >= 1000");   // helper.go:166-171
        o.validate_len = (uint64_t)vl;
        if (g.alphabet != Auto && g.alphabet != Unlimit) {                  // seq.go:90-93
            fprintf(stderr, "[INFO] when flag -t (--seq-type) given, flag -v (--validate-seq) is automatically switched on\n");
            o.validate = true;
        }
        if (o.lower_case && o.upper_case)                                   // seq.go:96-98
            return fatal("could not give both flags -l (--lower-case) and -u (--upper-case)");
        return cmd_seq(g, o, files);
    }
    if (is_cv) {
        ConvertOpts o;
        auto encoding = [](std::string v) -> int {                          // convert.go:293-311
            for (auto& ch : v) ch = (char)tolower((unsigned char)ch);
            return v == "sanger" ? SKH_Q_SANGER : v == "solexa" ? SKH_Q_SOLEXA : v == "illumina-1.3+" ? SKH_Q_ILLUMINA13
                 : v == "illumina-1.5+" ? SKH_Q_ILLUMINA15 : v == "illumina-1.8+" ? SKH_Q_ILLUMINA18 : v.empty() ? 0 : -1;
        };
        o.from = encoding(p.value("from", ""));
        if (o.from < 0) return fatal("unsupported quality encoding: " + p.value("from", ""));
        o.to = encoding(p.value("to", "Sanger"));
        if (o.to < 0) return fatal("unsupported quality encoding: " + p.value("to", ""));
        o.dry_run = p.has("dry-run"); o.force = p.has("force");
        o.nrecords = atoi(p.value("nrecords", "1000").c_str());
        if (o.nrecords <= 0) return fatal("value of flag --nrecords should be greater than 0");
        o.n_most_common = atoi(p.value("thresh-B-in-n-most-common", "4").c_str());
        if (o.n_most_common <= 0) return fatal("value of flag --thresh-B-in-n-most-common should be greater than 0");
        o.illumina15_frac = atof(p.value("thresh-illumina1.5-frac", "0.1").c_str());
        return cmd_convert(g, o, files);
    }
    if (is_sub) {
        SubseqOpts o;
        o.region = p.value("region", ""); o.gtf = p.value("gtf", ""); o.bed = p.value("bed", "");
        o.gtf_tag = p.value("gtf-tag", "gene_id");
        o.chr = p.values("chr"); o.feature = p.values("feature");
        o.up_stream = atoi(p.value("up-stream", "0").c_str());
        o.down_stream = atoi(p.value("down-stream", "0").c_str());
        if (o.up_stream < 0) return fatal("value of flag --up-stream should be greater than 0");
        if (o.down_stream < 0) return fatal("value of flag --down-stream should be greater than 0");
        o.only_flank = p.has("only-flank");
        return cmd_subseq(g, o, files);
    }
    if (is_fx) {
        Fx2tabOpts o;
        o.only_id = p.has("only-id"); o.length = p.has("length"); o.gc = p.has("gc"); o.gc_skew = p.has("gc-skew");
        o.only_name = p.has("name"); o.header_line = p.has("header-line"); o.alphabet_col = p.has("alphabet");
        o.base = p.values("base-content");
        return cmd_fx2tab(g, o, files);
    }
    StatsOpts o;
    o.all = p.has("all"); o.tabular = p.has("tabular");
    o.gap_letters = p.value("gap-letters", "- .");
    return cmd_stats(g, o, files);
}
