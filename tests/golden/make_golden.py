#!/usr/bin/env python3
"""Extract the reference's documented known answers for the hot path into
tests/golden/expected/ (run once in the build container; the result is committed).

Every vector is a worked example printed in the reference's own docs, i.e. output
of the real Go binary as its authors recorded it.  The inputs are the fixture
files of the reference's tests/ directory, copied verbatim next to this script.
Only data is taken: line ranges of documentation, never source code.

  python tests/golden/make_golden.py /root/reference
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (doc file, first line, last line (1-based, inclusive), indent to strip,
#          argv for the CLI relative to tests/golden, comparison mode, note)
VECTORS = {
    "G1_seq_revcomp_head": ("doc/docs/usage.md", 364, 366, 8,
        ["seq", "hairpin.fa.gz", "-r", "-p"], "head3",
        "first record of `seqkit seq hairpin.fa.gz -r -p` (RNA complement table, 60-col wrap)"),
    "G1b_seq_plain_head": ("doc/docs/usage.md", 284, 286, 12,
        ["seq", "hairpin.fa.gz"], "head3", "reader + wrap, first record"),
    "G1c_seq_s_w0_head": ("doc/docs/usage.md", 353, 355, 8,
        ["seq", "hairpin.fa.gz", "-s", "-w", "0"], "head3", "-s -w 0, first three records"),
    "G1d_seq_names_head": ("doc/docs/usage.md", 332, 334, 12,
        ["seq", "hairpin.fa.gz", "-n"], "head3", "full names"),
    "G1e_seq_ids_head": ("doc/docs/usage.md", 339, 341, 12,
        ["seq", "hairpin.fa.gz", "-n", "-i"], "head3", "IDs"),
    "G1f_seq_custom_ids_head": ("doc/docs/usage.md", 346, 348, 12,
        ["seq", "hairpin.fa.gz", "-n", "-i", "--id-regexp", "^[^\\s]+\\s([^\\s]+)\\s"], "head3",
        "IDs cut out by a custom --id-regexp (capture group 1)"),
    "G2_fx2tab_len_gc_head": ("doc/docs/usage.md", 795, 798, 8,
        ["fx2tab", "hairpin.fa.gz", "-l", "-g", "-n", "-i", "-H"], "fields4",
        "piped through `csvtk pretty` in the doc: compare whitespace-split fields of the first 4 lines"),
    "G3_stats_pretty": ("doc/docs/usage.md", 607, 611, 8,
        ["stats", "hairpin.fa.gz", "mature.fa.gz", "reads_1.fq.gz", "reads_2.fq.gz"], "exact",
        "pretty table, 4 files"),
    "G3_stats_tabular": ("doc/docs/usage.md", 616, 621, 8,
        ["stats", "hairpin.fa.gz", "mature.fa.gz", "Illimina1.8.fq.gz", "reads_1.fq.gz",
         "reads_2.fq.gz", "-T"], "expandtabs",
        "-T output; the doc shows tabs expanded to 8-column stops"),
    "G3_stats_all": ("doc/docs/usage.md", 656, 661, 8,
        ["stats", "hairpin.fa.gz", "mature.fa.gz", "Illimina1.8.fq.gz", "reads_1.fq.gz",
         "reads_2.fq.gz", "-a"], "exact", "-a table: quartiles, sum_gap, N50"),
    "G4_stats_dna": ("doc/docs/usage.md", 304, 305, 12,
        ["stats"], "exact", "stdin: >seq\\nacgtryswkmbdhvACGTRYSWKMBDHV"),
    "G4_stats_rna": ("doc/docs/usage.md", 308, 309, 12,
        ["stats"], "exact", "stdin: >seq\\nACGUN ACGUN  (interior space kept: sum_len 11)"),
    "G4_stats_protein": ("doc/docs/usage.md", 312, 313, 12,
        ["stats"], "exact", "stdin: >seq\\nabcdefghijklmnpqrstvwyz"),
    "G4_stats_fastq": ("doc/docs/usage.md", 316, 317, 12,
        ["stats"], "exact", "stdin: @read\\nACTGCN\\n+\\n@IICCG"),
    "G9_seq_illumina15": ("doc/docs/usage.md", 944, 947, 0,
        ["seq", "Illimina1.5.fq"], "exact", "FASTQ round trip, '+' line reduced to '+'"),
    "G10_seq_validate": ("doc/docs/usage.md", 323, 324, 12,
        ["seq", "-t", "dna"], "stderr255",
        "sequence validation switched on by -t: both stderr lines, nothing on stdout, exit status 255"),
}

STDIN = {
    "G4_stats_dna": ">seq\nacgtryswkmbdhvACGTRYSWKMBDHV\n",
    "G4_stats_rna": ">seq\nACGUN ACGUN\n",
    "G4_stats_protein": ">seq\nabcdefghijklmnpqrstvwyz\n",
    "G4_stats_fastq": "@read\nACTGCN\n+\n@IICCG\n",
    "G10_seq_validate": ">seq\nabcdefghijklmnpqrstvwyz\n",
}


def unescape(t):
    return t.replace("\\t", "\t").replace("\\n", "\n")


def subseq_vectors(ref, out_dir, manifest):
    """subseq known answers: the assertions of tests/test.sh:104-166 (every command there is
    piped through `seq -s -w 0`) and the worked examples of doc/docs/usage.md:408-419, 455-496.
    The small GTF / FASTA inputs those lines spell out are written next to this script."""
    import re
    with open(os.path.join(ref, "tests/test.sh"), encoding="utf-8") as fh:
        sh = fh.read().split("\n")
    with open(os.path.join(ref, "doc/docs/usage.md"), encoding="utf-8") as fh:
        md = fh.read().split("\n")
    pipe = ["seq", "-s", "-w", "0"]

    def add(name, source, argv, stdin, want, note):
        with open(os.path.join(out_dir, name + ".txt"), "w", encoding="utf-8") as fh:
            fh.write(want)
        manifest[name] = {"source": source, "argv": argv, "pipe": pipe, "mode": "exact", "note": note, "stdin": stdin}

    # test.sh:105  echo -e ">seq\nacgtnACGTN"
    testseq = unescape(re.search(r'echo -e "([^"]*)"', sh[104]).group(1)) + "\n"
    for cmd_line, assert_line in ((110, 113), (116, 119), (122, 125), (128, 131), (134, 137)):
        region = re.search(r"subseq -r (\S+)", sh[cmd_line - 1]).group(1)
        want = re.search(r"assert_equal (\S+) ", sh[assert_line - 1]).group(1) + "\n"
        add("G11_subseq_region_%d" % cmd_line, "tests/test.sh:%d,%d" % (cmd_line, assert_line),
            ["subseq", "-r", region], testseq, want, "region " + region)
    # test.sh:142  gtf="..." (two CDS features on 4..6, one per strand)
    gtf = unescape(re.search(r'gtf="(.*)"$', sh[141]).group(1).replace('\\"', '"'))
    with open(os.path.join(HERE, "subseq_test.gtf"), "w") as fh:
        fh.write(gtf)
    for cmd_line, assert_line in ((145, 148), (151, 154), (157, 160), (163, 166)):
        flags = re.search(r"--gtf <\(echo -ne \$gtf\)(.*?) \|", sh[cmd_line - 1]).group(1).split()
        want = unescape(re.search(r'echo -e "([^"]*)"', sh[assert_line - 1]).group(1)) + "\n"
        add("G12_subseq_gtf_%d" % cmd_line, "tests/test.sh:%d,%d (md5 of this text)" % (cmd_line, assert_line),
            ["subseq", "--gtf", "subseq_test.gtf"] + flags, testseq, want, "GTF features, flags " + " ".join(flags))
    # usage.md:410-419  region table on ACGTNacgtn
    row = md[409].split()
    assert row[0] == "seq"
    table_seq = ">seq\n" + "".join(row[1:]) + "\n"
    for ln in range(411, 420):
        f = md[ln - 1].split()
        add("G13_subseq_table_%d" % ln, "doc/docs/usage.md:410,%d" % ln, ["subseq", "-r", f[0]], table_seq,
            "".join(f[1:]) + "\n", "region table row " + f[0])
    # usage.md:455-467, 484-496: t.fa, t.gtf and three outputs.  The documented head lines use an older
    # name layout (seq_5:8:.) than the code of this tree (seq_5-8:., subseq.go:453), so only the
    # sequences are compared.
    with open(os.path.join(HERE, "subseq_doc.fa"), "w") as fh:
        fh.write(md[455].strip() + "\n" + md[456].strip() + "\n")
    with open(os.path.join(HERE, "subseq_doc.gtf"), "w") as fh:
        for ln in (460, 461):
            f = md[ln - 1].split()
            fh.write("\t".join(f[:8]) + "\t" + " ".join(f[8:]) + "\n")
    for name, cmd_line, rows in (("G14_subseq_doc_gtf", 463, (465, 467)), ("G14_subseq_doc_gtf_u3", 484, (486, 488)),
                                 ("G14_subseq_doc_gtf_u3f", 492, (494, 496))):
        flags = md[cmd_line - 1].split("t.fa")[1].split()
        add(name, "doc/docs/usage.md:%d-%d" % (cmd_line, rows[1]), ["subseq", "--gtf", "subseq_doc.gtf", "subseq_doc.fa"] + flags,
            None, "".join(md[r - 1].strip() + "\n" for r in rows), "documented GTF example, sequences only")


def convert_vectors(ref, out_dir, manifest):
    """convert known answers: the worked examples of doc/docs/usage.md:878-956.  Each block shows the
    log lines on stderr ("[INFO] ...") followed by the first record (`| seqkit head -n 1`)."""
    with open(os.path.join(ref, "doc/docs/usage.md"), encoding="utf-8") as fh:
        md = fh.read().split("\n")
    cases = [("G15_convert_default", 879, 886, ["convert", "Illimina1.8.fq.gz"], None),
             ("G15_convert_force", 893, 899, ["convert", "Illimina1.8.fq.gz", "-f"], None),
             ("G15_convert_to_illumina15", 908, 914, ["convert", "Illimina1.8.fq.gz", "--to", "Illumina-1.5+"], None),
             # the second command's log lines and output; the first command's are the vector above
             ("G15_convert_round_trip", 924, 930, ["convert", "Illimina1.8.fq.gz", "--to", "Illumina-1.5+"], ["convert"]),
             ("G15_convert_illumina15_file", 950, 956, ["convert", "Illimina1.5.fq"], None)]
    for name, a, b, argv, pipe in cases:
        with open(os.path.join(out_dir, name + ".txt"), "w", encoding="utf-8") as fh:
            fh.write("".join(l + "\n" for l in md[a - 1:b]))
        manifest[name] = {"source": "doc/docs/usage.md:%d-%d" % (a, b), "argv": argv, "mode": "log+head4", "stdin": None,
                          "note": "stderr lines, then the first record"}
        if pipe:
            manifest[name]["pipe"] = pipe
    with open(os.path.join(out_dir, "G15_convert_from_solexa.txt"), "w", encoding="utf-8") as fh:
        fh.write("".join(l + "\n" for l in md[936:938]))
    manifest["G15_convert_from_solexa"] = {"source": "doc/docs/usage.md:937-938", "mode": "stderr255", "stdin": None,
                                           "argv": ["convert", "Illimina1.8.fq.gz", "--from", "Solexa"],
                                           "note": "a byte below the Solexa minimum: both stderr lines, exit status 255"}


def more_vectors(ref, out_dir, manifest):
    """Further worked examples on the fixture files: the length filters of `seq` seen through `stats`
    (doc/docs/usage.md:382-392; the documented commands read hairpin.fa from stdin, the file name
    printed is that of the piped `stats`), the default rows of `fx2tab` (usage.md:786-788, tabs shown
    expanded) and `stat`, the alias of `stats` (doc/docs/tutorial.md:237-239)."""
    with open(os.path.join(ref, "doc/docs/usage.md"), encoding="utf-8") as fh:
        md = fh.read().split("\n")
    with open(os.path.join(ref, "doc/docs/tutorial.md"), encoding="utf-8") as fh:
        tut = fh.read().split("\n")

    def put(name, lines, indent, source, argv, mode, note, pipe=None):
        with open(os.path.join(out_dir, name + ".txt"), "w", encoding="utf-8") as fh:
            fh.write("".join(l[indent:].rstrip() + "\n" for l in lines))
        manifest[name] = {"source": source, "argv": argv, "mode": mode, "note": note, "stdin": None}
        if pipe:
            manifest[name]["pipe"] = pipe

    for name, cmd_line, flags in (("G16_seq_len_filter_none", 382, []), ("G16_seq_len_filter_m100", 386, ["-m", "100"]),
                                  ("G16_seq_len_filter_m100_M1000", 390, ["-m", "100", "-M", "1000"])):
        assert md[cmd_line - 1].split("seqkit seq")[1].split("|")[0].split() == flags
        put(name, md[cmd_line:cmd_line + 2], 8, "doc/docs/usage.md:%d-%d" % (cmd_line, cmd_line + 2),
            ["seq", "hairpin.fa.gz"] + flags, "exact", "seq %s | stats" % " ".join(flags), pipe=["stats"])
    put("G17_fx2tab_default_rows", md[786:788], 8, "doc/docs/usage.md:786-788", ["fx2tab", "hairpin.fa.gz"],
        "head2_expandtabs", "name, sequence and (empty) quality columns of the first two records")
    # the transforms behind reverse / complement in the record loop of seq (seq.go:196-241)
    for name, first, stdin, argv in (("G19_seq_remove_gaps_upper", 371, ">seq\nACGT-ACTGC-ACC\n", ["seq", "-g", "-u"]),
                                     ("G19_seq_rna2dna", 377, ">seq\nUCAUAUGCUUGUCUCAAAGAUUA\n", ["seq", "--rna2dna"])):
        assert md[first - 2].strip().endswith("seqkit " + " ".join(argv)), md[first - 2]
        put(name, md[first - 1:first + 1], 8, "doc/docs/usage.md:%d-%d" % (first - 1, first + 1), argv, "exact",
            "stdin: " + stdin.replace("\n", "\\n"))
        manifest[name]["stdin"] = stdin
    put("G18_stat_alias", tut[237:239], 8, "doc/docs/tutorial.md:237-239", ["stat", "hairpin.fa.gz"], "exact",
        "`stat` is `stats`")


def main(ref):
    out_dir = os.path.join(HERE, "expected")
    os.makedirs(out_dir, exist_ok=True)
    manifest = {}
    for name, (doc, a, b, indent, argv, mode, note) in VECTORS.items():
        with open(os.path.join(ref, doc), encoding="utf-8") as fh:
            lines = fh.read().split("\n")[a - 1:b]
        text = "".join(l[indent:] + "\n" for l in lines)
        with open(os.path.join(out_dir, name + ".txt"), "w", encoding="utf-8") as fh:
            fh.write(text)
        manifest[name] = {"source": "%s:%d-%d" % (doc, a, b), "argv": argv, "mode": mode,
                          "note": note, "stdin": STDIN.get(name)}
    # G3b: per-record lengths of hairpin.fa from the reference's own .fai (key, length, ...)
    lens = []
    with open(os.path.join(ref, "tests/hairpin.fa.fai")) as fh:
        for l in fh:
            f = l.rstrip("\n").split("\t")
            lens.append(int(f[1]))
    with open(os.path.join(out_dir, "G3b_hairpin_lengths.txt"), "w") as fh:
        fh.write("\n".join(map(str, lens)) + "\n")
    manifest["G3b_hairpin_lengths"] = {"source": "tests/hairpin.fa.fai column 2", "mode": "lengths",
                                       "argv": ["fx2tab", "hairpin.fa.gz", "-n", "-l"], "stdin": None,
                                       "note": "record lengths in file order"}
    subseq_vectors(ref, out_dir, manifest)
    convert_vectors(ref, out_dir, manifest)
    more_vectors(ref, out_dir, manifest)
    with open(os.path.join(out_dir, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote %d vectors to %s" % (len(manifest), out_dir))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
