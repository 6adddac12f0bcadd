#!/usr/bin/env python3
"""Serialises the known-answer vectors of the reference's own unit tests into tests/golden/*.json.

The reference (PacificBiosciences/HiPhase v1.5.0) is Rust and cannot be built or imported here, so the
inputs and expected outputs below are transcribed BY HAND from its `#[cfg(test)]` modules; each entry
cites the file:line it comes from. Data only — no reference source is reproduced.
Run:  python tests/golden/make_golden.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")


# ------------------------------------------------------------------------------------------------
# src/data_types/read_segments.rs:213-308
# ------------------------------------------------------------------------------------------------
Q = [0, 1, 2, 3, 4, 5, 6, 7, 0, 0]
read_segments = {
    "source": "src/data_types/read_segments.rs:213-308",
    "constructor": {  # test_constructor :214-227
        "alleles": [3, 0, 1, 0, 0, 1, 2, 2, 3, 3], "quals": Q,
        "expect_alleles": [0, 1, 0, 0, 1], "expect_quals": [1, 2, 3, 4, 5], "expect_region": [1, 6],
    },
    "score_haplotype": {  # test_score_haplotype :230-250
        "alleles": [3, 0, 1, 0, 0, 1, 2, 1, 3, 3], "quals": Q, "expect_region": [1, 8], "expect_num_set": 6,
        "cases": [
            {"haplotype": [0, 0, 1, 0, 0, 1, 1, 1, 0, 0], "offset": 0, "expect": 6},
            {"haplotype": [2] * 10, "offset": 0, "expect": 0},
            {"haplotype": [1, 1, 0, 1, 1, 0, 0, 0, 1, 1], "offset": 0, "expect": 28},
        ],
    },
    "score_partial_haplotype": {  # test_score_partial_haplotype :253-275
        "alleles": [2, 0, 1, 0, 0, 1, 2, 1, 2, 2], "quals": Q,
        "cases": (
            [
                {"haplotype": [0, 1, 0, 0, 1, 1, 1], "offset": 1, "expect": 6},
                {"haplotype": [2] * 7, "offset": 2, "expect": 0},
                {"haplotype": [1, 0, 1, 1, 0, 0, 0], "offset": 1, "expect": 28},
            ]
            + [  # for x in 0..7: score(&haplotype[x..], 1+x) == sum((x+1)..8)
                {"haplotype": [1, 0, 1, 1, 0, 0, 0][x:], "offset": 1 + x, "expect": sum(range(x + 1, 8))}
                for x in range(7)
            ]
        ),
    },
    "collapse": {  # test_collapse :278-308
        "rows": [
            {"alleles": [3, 1, 0, 2, 1, 3, 3], "quals": [0, 2, 1, 0, 2, 0, 0]},
            {"alleles": [3, 3, 0, 1, 0, 1, 1], "quals": [0, 0, 1, 2, 2, 1, 1]},
        ],
        "expect_alleles": [3, 1, 0, 2, 2, 1, 1], "expect_quals": [0, 2, 1, 0, 0, 1, 1], "expect_region": [1, 7],
        "haplotype": [0, 1, 0, 0, 0, 1, 0], "expect_score": 1,
    },
}
dump("read_segments.json", read_segments)

# ------------------------------------------------------------------------------------------------
# src/astar_phaser.rs:636-798
# ------------------------------------------------------------------------------------------------
n = 4
heur = [n - i for i in range(n + 1)]
astar = {
    "source": "src/astar_phaser.rs:636-798",
    "astarnode": {  # test_astarnode :663-766; get_simple_reads :642-660
        "reads": [
            {"alleles": [0] * n, "quals": [2] * n},
            {"alleles": [1] * n, "quals": [3] * n},
        ],
        "heuristic_costs": heur, "hap_offset": 0,
        "walks": [
            {"name": "all 0-hom", "path1": [0] * n, "path2": [0] * n,
             "expect_total": [heur[i + 1] + 3 * (i + 1) for i in range(n)],
             "expect_frozen": [0, 0, 0, 3 * n], "expect_hets": [0] * n},
            {"name": "all het 0|1", "path1": [0] * n, "path2": [1] * n,
             "expect_total": [heur[i + 1] for i in range(n)],
             "expect_frozen": [0] * n, "expect_hets": [1, 2, 3, 4]},
            {"name": "all 1-hom", "path1": [1] * n, "path2": [1] * n,
             "expect_total": [heur[i + 1] + 2 * (i + 1) for i in range(n)],
             "expect_frozen": [0, 0, 0, 2 * n], "expect_hets": [0] * n},
        ],
    },
    "pqueuehaptracker": {  # test_pqueuehaptracker :769-798 ; ops: 0 add, 1 remove, 2 increase_threshold
        "max_hap_length": 10,
        "script": (
            [[0, i, i + 1] for i in range(11)]          # add 0..10 -> len 1..11
            + [[1, 3, 10]]                               # remove 3 -> 10
            + [[2, 4, 7]]                                # threshold 4 -> 7
            + [[1, i, 7] for i in range(3)]              # removing below threshold keeps 7
            + [[0, 0, 7]]                                # adding below threshold keeps 7
            + [[2, 4, 7]]                                # same threshold is a no-op
        ),
    },
}
dump("astar_phaser.json", astar)

# ------------------------------------------------------------------------------------------------
# src/phaser.rs:756-804
# ------------------------------------------------------------------------------------------------
phaser = {
    "source": "src/phaser.rs:756-804",
    "span_counts": {  # test_get_solution_span_counts :757-775
        "h1": [0, 1, 1, 0, 0, 0], "h2": [1, 1, 1, 1, 0, 1],
        "reads": [
            {"name": "r1", "alleles": [0, 0, 0, 0, 0, 0], "quals": [1, 1, 1, 1, 1, 1]},
            {"name": "r2", "alleles": [3, 3, 3, 1, 1, 3], "quals": [0, 0, 0, 1, 1, 0]},
            {"name": "r3", "alleles": [1, 1, 1, 1, 3, 3], "quals": [1, 1, 1, 1, 0, 0]},
            {"name": "r4", "alleles": [3, 1, 1, 1, 1, 1], "quals": [0, 1, 1, 1, 1, 1]},
        ],
        "expect": [2, 2, 2, 2, 2],
    },
    "haplotag": {  # test_haplotag_reads :778-804
        "h1": [0] * 6, "h2": [1] * 6, "block_tags": [0, 0, 0, 3, 3, 5],
        "reads": [
            {"name": "r1", "alleles": [0, 0, 0, 0, 0, 0], "quals": [1, 1, 1, 1, 1, 1]},
            {"name": "r2", "alleles": [2, 2, 2, 1, 1, 2], "quals": [0, 0, 0, 1, 1, 0]},
            {"name": "r3", "alleles": [2, 2, 2, 1, 0, 2], "quals": [0, 0, 0, 1, 1, 0]},
            {"name": "r4", "alleles": [2, 2, 2, 1, 0, 1], "quals": [0, 0, 0, 1, 1, 1]},
            {"name": "r5", "alleles": [2, 2, 2, 1, 0, 2], "quals": [0, 0, 0, 2, 1, 0]},
        ],
        # name -> [phase block id, haplotag]; null = not tagged. r5 is not asserted upstream; its value
        # follows from the same rule (score h1 = 2 > score h2 = 1 -> haplotag 1, first het the read resolves = 3).
        "expect": {"r1": [0, 0], "r2": [3, 1], "r3": None, "r4": [3, 1], "r5": [3, 1]},
    },
}
dump("phaser.json", phaser)

# ------------------------------------------------------------------------------------------------
# src/sequence_alignment.rs:40-76 and src/data_types/variants.rs:799-846
# ------------------------------------------------------------------------------------------------
v1, v2, v3, v4 = [0, 1, 2, 4, 5], [0, 1, 3, 4, 5], [1, 2, 3, 5], []
e1 = [65] * 17 + [67] + [65] * 3
e2 = [65] * 10 + [67] + [65] * 6 + [67] + [65] * 3
e3 = [65] * 16 + [67] + [65] * 3
seq_align = {
    "source": "src/sequence_alignment.rs:44-76; src/data_types/variants.rs:799-846",
    "edit_distance": [  # test_edit_distance :45-64
        [v1, v1, 0], [v1, v2, 1], [v1, v3, 2], [v1, v4, 5],
        [v2, v2, 0], [v2, v3, 3], [v2, v4, 5],
        [v3, v3, 0], [v3, v4, 4], [v4, v4, 0],
        # test_edit_error_001 :67-76
        [e1, e3, 1], [e2, e3, 1], [e3, e1, 1], [e3, e2, 1],
    ],
    # test_reference_adjustment :800-846 — new_indel(0,20,2,"A","AGT",1,2) + prefix "AC" + postfix "GGCC",
    # then truncate_reference_postfix(1): padded allele0 = "ACAGGC", allele1 = "ACAGTGGC", prefix 2, postfix 3.
    "closest_allele": {
        "allele0": "ACAGGC", "allele1": "ACAGTGGC", "prefix_len": 2, "postfix_len": 3,
        "truncated_allele0": "A", "truncated_allele1": "AGT",
        "cases": [  # (observed, expected AlleleType, min distance, other distance)
            ["A", 0, 5, 7], ["AGT", 0, 4, 5], ["AG", 0, 4, 6],
            ["ACAGGC", 0, 0, 2], ["ACAGTGGC", 1, 0, 2], ["ACAGGGC", 2, 1, 1],
        ],
        # same test, :832-835: the un-padded alleles no longer match exactly once prefix/postfix are attached
        "match_after_padding": [["A", 2], ["AGT", 2], ["AG", 2]],
    },
    # Variant::match_allele on the constructor tests, variants.rs:668-797: (kind, position, ref_len, allele0,
    # allele1, index_allele0, index_allele1, [(observed, expected 0/1/2)])
    "match_allele": [
        ["snv", 1, 1, "A", "C", 0, 1, [["A", 0], ["C", 1], ["G", 2], ["T", 2]]],                  # :668-687
        ["deletion", 10, 3, "AGT", "A", 0, 1, [["AGT", 0], ["A", 1], ["AG", 2]]],                # :689-700
        ["deletion", 10, 4, "C", "A", 1, 2, [["ACCC", 2], ["C", 0], ["A", 1]]],                  # :702-714
        ["insertion", 20, 1, "A", "AGT", 0, 1, [["A", 0], ["AGT", 1], ["AG", 2]]],               # :718-731
        ["indel", 20, 2, "A", "AGT", 1, 2, [["A", 0], ["AGT", 1], ["AG", 2]]],                   # :733-747
        ["sv_insertion", 20, 1, "A", "AGT", 0, 1, [["A", 0], ["AGT", 1], ["AG", 2]]],            # :749-763
        ["sv_deletion", 10, 3, "AGT", "A", 0, 1, [["AGT", 0], ["A", 1], ["AG", 2]]],             # :765-780
        ["tandem_repeat", 10, 4, "AAAC", "AAACAAAC", 0, 1, [["AAAC", 0], ["AAACAAAC", 1], ["AAACAA", 2]]],  # :782-797
    ],
}
dump("sequence_alignment.json", seq_align)

# ------------------------------------------------------------------------------------------------
# src/wfa_graph.rs:672-1208
# ------------------------------------------------------------------------------------------------
def q(seq, score, nodes=None):
    if isinstance(seq, str):
        seq = list(seq.encode())
    return {"seq": seq, "score": score, "nodes": nodes}


base = [0, 1, 2, 4, 5]
hand = [
    {"name": "test_single_node", "line": 677,
     "nodes": [{"seq": base, "parents": []}],
     "queries": [q(base, 0, [0]), q([0, 1, 3, 4, 5], 1), q([1, 2, 3, 5], 2), q([], 5)]},
]
for sp in range(len(base)):  # test_two_node_single_path :696-716
    hand.append({"name": f"test_two_node_single_path[{sp}]", "line": 696,
                 "nodes": [{"seq": base[:sp], "parents": []}, {"seq": base[sp:], "parents": [0]}],
                 "queries": [q(base, 0, [0, 1]), q([0, 1, 3, 4, 5], 1, [0, 1]), q([1, 2, 3, 5], 2, [0, 1]),
                             q([], 5, [0, 1])]})
hand.append({"name": "test_basic_variant", "line": 719,
             "nodes": [{"seq": [0, 1], "parents": []}, {"seq": [2], "parents": [0]}, {"seq": [3], "parents": [0]},
                       {"seq": [4, 5], "parents": [1, 2]}],
             "queries": [q(base, 0, [0, 1, 3]), q([0, 1, 3, 4, 5], 0, [0, 2, 3]), q([1, 2, 3, 5], 2, [0, 1, 3]),
                         q([], 5, [0, 1, 2, 3]), q([0, 1, 4, 5], 1, [0, 1, 2, 3])]})
t1, t2, t3 = [0, 1, 2, 3, 4, 5], [0, 1, 2, 4, 4, 5], [0, 1, 4, 4, 5]
hand.append({"name": "test_triple_split", "line": 747,
             "nodes": [{"seq": [0, 1], "parents": []}, {"seq": [2, 3], "parents": [0]}, {"seq": [2, 4], "parents": [0]},
                       {"seq": [4], "parents": [0]}, {"seq": [4, 5], "parents": [1, 2, 3]}],
             "queries": [q(t1, 0, [0, 1, 4]), q(t2, 0, [0, 2, 4]), q(t3, 0, [0, 3, 4])]})
hand.append({"name": "test_nested_split", "line": 766,
             "nodes": [{"seq": [0, 1], "parents": []}, {"seq": [2, 3], "parents": [0]}, {"seq": [2], "parents": [0]},
                       {"seq": [4], "parents": [0, 2]}, {"seq": [4, 5], "parents": [1, 3]}],
             "queries": [q(t1, 0, [0, 1, 4]), q(t2, 0, [0, 2, 3, 4]), q(t3, 0, [0, 3, 4])]})
hand.append({"name": "test_double_split", "line": 789,
             "nodes": [{"seq": [0, 1], "parents": []}, {"seq": [2], "parents": [0]}, {"seq": [], "parents": [0, 1]},
                       {"seq": [3], "parents": [2]}, {"seq": [4], "parents": [2]}, {"seq": [4, 5], "parents": [3, 4]}],
             "queries": [q(t1, 0, [0, 1, 2, 3, 5]), q(t2, 0, [0, 1, 2, 4, 5]), q(t3, 0, [0, 2, 4, 5])]})
hand.append({"name": "test_overlapping_split", "line": 814,
             "nodes": [{"seq": [0], "parents": []}, {"seq": [1], "parents": [0]}, {"seq": [2], "parents": [1]},
                       {"seq": [3], "parents": [0, 2]}, {"seq": [4, 5], "parents": [1, 3]}],
             "queries": [q([0, 1, 2, 3, 4, 5], 0, [0, 1, 2, 3, 4]), q([0, 3, 4, 5], 0, [0, 3, 4]),
                         q([0, 1, 4, 5], 0, [0, 1, 4])]})


def var(kind, **kw):
    d = {"kind": kind}
    d.update(kw)
    return d


def snv(pos, a0, a1, i0=0, i1=1, vcf=0):
    return var("snv", vcf_index=vcf, position=pos, allele0=a0, allele1=a1, index_allele0=i0, index_allele1=i1)


def dele(pos, ref_len, a0, a1, i0=0, i1=1):
    return var("deletion", vcf_index=0, position=pos, ref_len=ref_len, allele0=a0, allele1=a1, index_allele0=i0,
               index_allele1=i1)


def ins(pos, a0, a1, i0=0, i1=1, vcf=0):
    return var("insertion", vcf_index=vcf, position=pos, allele0=a0, allele1=a1, index_allele0=i0, index_allele1=i1)


def indel(pos, ref_len, a0, a1, i0, i1):
    return var("indel", vcf_index=0, position=pos, ref_len=ref_len, allele0=a0, allele1=a1, index_allele0=i0,
               index_allele1=i1)


built = [
    {"name": "test_simple_snv", "line": 842, "reference": "AAA", "ref_start": 0, "ref_end": 3,
     "variants": [snv(1, "A", "C")], "homs": [], "num_nodes": 4,
     "queries": [q("AAA", 0, [0, 2, 3]), q("ACA", 0, [0, 1, 3]), q("AA", 1, [0, 1, 2, 3])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[0, 0]], "3": []}},
    {"name": "test_multiple_variants", "line": 864, "reference": "AAAAA", "ref_start": 0, "ref_end": 5,
     "variants": [snv(1, "A", "C"), snv(3, "A", "C")], "homs": [], "num_nodes": 7,
     "queries": [q("AAAAA", 0, [0, 2, 3, 5, 6]), q("ACAAA", 0, [0, 1, 3, 5, 6]), q("AAACA", 0, [0, 2, 3, 4, 6]),
                 q("ACACA", 0, [0, 1, 3, 4, 6]), q("AAA", 2, [0, 1, 2, 3, 4, 5, 6]), q("AGAGA", 2, [0, 1, 2, 3, 4, 5, 6]),
                 q("GAAAA", 1, [0, 2, 3, 5, 6]), q("ACAGAA", 1, [0, 1, 3, 5, 6])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[0, 0]], "3": [], "4": [[1, 1]], "5": [[1, 0]], "6": []}},
    {"name": "test_overlapping_variants", "line": 905, "reference": "ACGTA", "ref_start": 0, "ref_end": 5,
     "variants": [dele(1, 2, "CG", "C"), dele(2, 2, "GT", "G")], "homs": [], "num_nodes": 7,
     "queries": [q("ACGTA", 0, [0, 2, 4, 5, 6]), q("ACTA", 0, [0, 1, 5, 6]), q("ACGA", 0, [0, 2, 3, 6]),
                 q("AGTA", 1, [0, 1, 2, 4, 5, 6]), q("AA", 2, [0, 1, 2, 3, 5, 6])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[0, 0]], "3": [[1, 1]], "4": [[1, 0]], "5": [], "6": []}},
    {"name": "test_identical_insertions", "line": 943, "reference": "ACGTA", "ref_start": 0, "ref_end": 5,
     "variants": [ins(2, "G", "GT"), ins(2, "G", "GT", vcf=1)], "homs": [], "num_nodes": 5,
     "queries": [q("ACGTA", 0, [0, 3, 4]), q("ACGTTA", 0, [0, 1, 2, 4]), q("ACGATA", 1, [0, 1, 2, 3, 4])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[1, 1]], "3": [[0, 0], [1, 0]], "4": []}},
    {"name": "test_multiallelic_indel", "line": 976, "reference": "ACGTA", "ref_start": 0, "ref_end": 5,
     "variants": [indel(2, 2, "G", "GTT", 1, 2)], "homs": [], "num_nodes": 5,
     "queries": [q("ACGTA", 0, [0, 3, 4]), q("ACGA", 0, [0, 1, 4]), q("ACGTTA", 0, [0, 2, 4]),
                 q("ACGGA", 1, [0, 1, 3, 4]), q("ACGGTA", 1, [0, 2, 3, 4])],
     "node_to_alleles": {"0": [], "1": [[0, 0]], "2": [[0, 1]], "3": [], "4": []}},
    {"name": "test_partial_reference", "line": 1010, "reference": "AAAAAAA", "ref_start": 2, "ref_end": 5,
     "variants": [snv(3, "A", "C")], "homs": [], "num_nodes": 4,
     "queries": [q("AAA", 0, [0, 2, 3]), q("ACA", 0, [0, 1, 3]), q("AA", 1, [0, 1, 2, 3])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[0, 0]], "3": []}},
    {"name": "test_complex_problem", "line": 1036, "reference": "AACGTTGACGTCC", "ref_start": 2, "ref_end": 12,
     "variants": [dele(3, 4, "GTTG", "G"), dele(4, 2, "TT", "T"), snv(6, "A", "C", 1, 2)], "homs": [], "num_nodes": 9,
     "queries": [q("CGTTGACGTC", 0, [0, 2, 4, 7, 8]), q("CGACGTC", 0, [0, 1, 8]), q("CGTGACGTC", 0, [0, 2, 3, 7, 8]),
                 q("CGTTAACGTC", 0, [0, 2, 4, 5, 8]), q("CGTTCACGTC", 0, [0, 2, 4, 6, 8]),
                 q("CGTAACGTC", 0, [0, 2, 3, 5, 8]), q("CGTCACGTC", 0, [0, 2, 3, 6, 8]),
                 q("CGGACGTC", 1, [0, 1, 2, 3, 7, 8]), q("CGTACGTC", 1, [0, 1, 2, 3, 5, 6, 7, 8])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[0, 0]], "3": [[1, 1]], "4": [[1, 0]], "5": [[2, 0]],
                         "6": [[2, 1]], "7": [], "8": []}},
    {"name": "test_variant_before_start", "line": 1094, "reference": "NNNNNNNNNAACGTA", "ref_start": 10, "ref_end": 15,
     "variants": [snv(9, "A", "T"), snv(10, "A", "T")], "homs": [], "num_nodes": 4, "queries": [],
     "node_to_alleles": {"0": [], "1": [[1, 1]], "2": [[1, 0]], "3": []}},
    {"name": "test_span_ref_end", "line": 1120, "reference": "ACGTA", "ref_start": 0, "ref_end": 5,
     "variants": [dele(3, 3, "TAG", "T")], "homs": [], "num_nodes": 1, "queries": [],
     "node_to_alleles": {"0": []}},
    {"name": "test_hom_variants", "line": 1139, "reference": "AAAAA", "ref_start": 0, "ref_end": 5,
     "variants": [snv(3, "A", "C")], "homs": [snv(1, "A", "C")], "num_nodes": 7,
     "queries": [q("AAAAA", 0, [0, 2, 3, 5, 6]), q("ACAAA", 0, [0, 1, 3, 5, 6]), q("ACACA", 0, [0, 1, 3, 4, 6]),
                 q("ACAA", 1, [0, 1, 3, 4, 5, 6])],
     "node_to_alleles": {"0": [], "1": [], "2": [], "3": [], "4": [[0, 1]], "5": [[0, 0]], "6": []}},
    {"name": "test_variant_at_start", "line": 1167, "reference": "AAA", "ref_start": 0, "ref_end": 3,
     "variants": [snv(0, "A", "C")], "homs": [], "num_nodes": 4,
     "queries": [q("AAA", 0, [0, 2, 3]), q("CAA", 0, [0, 1, 3]), q("AA", 1, [0, 1, 2, 3])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[0, 0]], "3": []}},
    {"name": "test_variant_at_end", "line": 1189, "reference": "AAA", "ref_start": 0, "ref_end": 3,
     "variants": [snv(2, "A", "C")], "homs": [], "num_nodes": 4,
     "queries": [q("AAA", 0, [0, 2, 3]), q("AAC", 0, [0, 1, 3]), q("AA", 1, [0, 1, 2, 3])],
     "node_to_alleles": {"0": [], "1": [[0, 1]], "2": [[0, 0]], "3": []}},
]
dump("wfa_graph.json", {"source": "src/wfa_graph.rs:672-1208", "hand_built": hand, "variant_built": built})
print("golden vectors written to", HERE)
