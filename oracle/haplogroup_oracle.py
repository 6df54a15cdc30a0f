"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's haplogroup module around the
site pileup (find-y-branch / find-mt-branch), used by tests/ as the checker for
include/dut_haplogroup.h.  Nothing in the product imports this module.

Parity unpinned: the reference has no tests or fixtures for these functions and cannot be built
here; every function follows the cited lines of the reference source.

  parse_ftdna / parse_decodingus / build_tree   src/vendor/ftdna.rs:10-168, src/vendor/decoding_us.rs:10-220
  load_tree (root selection)                    src/haplogroup/tree.rs:27-47
  collect_snps, find_path_to_root               src/haplogroup/tree.rs:58-97
  validate_reference                            src/haplogroup/validation.rs:5-35
  call_sites                                    src/haplogroup/caller.rs:132-149
  calculate_haplogroup_score                    src/haplogroup/scoring.rs:8-148
  collect_scored_paths, get_snp_details, report src/haplogroup/mod.rs:92-258

HashMap-order dependent choices of the reference, fixed here the way include/dut_haplogroup.h
states them: DecodingUs children in ascending node index; rows that tie on (cumulative_snps, score)
in name order.
"""
import json

from . import report_oracle

CODE = "=ACMGRSVTWYHKDBN"


class TreeError(Exception):
    pass


def _u32(v):
    return isinstance(v, int) and not isinstance(v, bool) and 0 <= v <= 0xFFFFFFFF


def _i32(v):
    return isinstance(v, int) and not isinstance(v, bool) and -2 ** 31 <= v < 2 ** 31


def parse_ftdna(text):
    """-> all_nodes: {key: dict(haplogroup_id, parent_id, name, is_root, loci, children)}"""
    try:
        doc = json.loads(text, object_pairs_hook=lambda kv: dict(kv))
    except ValueError as e:
        raise TreeError("Failed to parse tree: " + str(e))
    if not isinstance(doc, dict) or not isinstance(doc.get("allNodes"), dict):
        raise TreeError("Failed to parse tree: missing field `allNodes`")
    all_nodes = {}
    for key, n in doc["allNodes"].items():
        if not isinstance(n, dict):
            raise TreeError("Failed to parse tree: FtdnaNode")
        for f, chk in (("haplogroupId", _u32), ("name", lambda v: isinstance(v, str)), ("isRoot", lambda v: isinstance(v, bool)),
                       ("root", lambda v: isinstance(v, str)), ("kitsCount", _u32), ("subBranches", _u32), ("bigYCount", _u32)):
            if f not in n or not chk(n[f]):
                raise TreeError("Failed to parse tree: field " + f)
        if "parentId" in n and not _u32(n["parentId"]):
            raise TreeError("Failed to parse tree: field parentId")
        loci = []
        variants = n.get("variants", [])
        if not isinstance(variants, list):
            raise TreeError("Failed to parse tree: field variants")
        for v in variants:
            if not isinstance(v, dict):
                raise TreeError("Failed to parse tree: FtdnaVariant")
            for f in ("variant", "ancestral", "derived", "region"):
                if f in v and not isinstance(v[f], str):
                    raise TreeError("Failed to parse tree: field " + f)
            if v.get("id") is not None and not _u32(v["id"]):
                raise TreeError("Failed to parse tree: field id")
            coords = {}
            pos = v.get("position")
            if pos is not None:
                if not _i32(pos):
                    raise TreeError("Failed to parse tree: field position")
                coords["GRCh38"] = dict(position=abs(pos), chromosome="chrY", ancestral=v.get("ancestral", ""), derived=v.get("derived", ""))
            loci.append(dict(name=v.get("variant", ""), snp=True, coordinates=coords))
        children = n.get("children", [])
        if not isinstance(children, list) or not all(_u32(c) for c in children):
            raise TreeError("Failed to parse tree: field children")
        all_nodes[key] = dict(haplogroup_id=n["haplogroupId"], parent_id=n.get("parentId", 0), name=n["name"], is_root=n["isRoot"],
                              loci=loci, children=list(children))
    return all_nodes


_ACC = {"CM000686.2": "GRCh38", "NC_000024.10": "GRCh38", "NC_060948.1": "T2T-CHM13v2.0", "CP086569.2": "T2T-CHM13v2.0",
        "CM000686.1": "GRCh37"}


def parse_decodingus(text):
    try:
        doc = json.loads(text)
    except ValueError as e:
        raise TreeError("Failed to parse tree: " + str(e))
    if not isinstance(doc, list):
        raise TreeError("Failed to parse tree: expected a sequence")
    for n in doc:
        ok = (isinstance(n, dict) and isinstance(n.get("name"), str) and isinstance(n.get("variants"), list)
              and isinstance(n.get("lastUpdated"), str) and isinstance(n.get("isBackbone"), bool)
              and (n.get("parentName") is None or isinstance(n.get("parentName"), str)))
        if not ok:
            raise TreeError("Failed to parse tree: ApiNode")
    name_to_id, root_id = {}, None
    for idx, n in enumerate(doc):
        name_to_id[n["name"]] = idx
        if not n.get("parentName"):
            if root_id is not None:
                raise TreeError("Failed to parse tree: Multiple root nodes found in tree")
            root_id = idx
    if root_id is None:
        raise TreeError("Failed to parse tree: No root node found")
    all_nodes = {}
    for idx, n in enumerate(doc):
        is_root = idx == root_id
        if is_root:
            parent_id = 0
        elif n.get("parentName"):
            parent_id = name_to_id.get(n["parentName"], root_id)
        else:
            parent_id = root_id
        loci = []
        for v in n["variants"]:
            if not (isinstance(v, dict) and isinstance(v.get("name"), str) and isinstance(v.get("variantType"), str)
                    and isinstance(v.get("coordinates"), dict)):
                raise TreeError("Failed to parse tree: ApiVariant")
            coords = {}
            for acc, c in v["coordinates"].items():
                if not (isinstance(c, dict) and _u32(c.get("start")) and _u32(c.get("stop")) and isinstance(c.get("anc"), str)
                        and isinstance(c.get("der"), str)):
                    raise TreeError("Failed to parse tree: ApiCoordinate")
                build = _ACC.get(acc, acc)
                coords[build] = dict(position=c["start"], chromosome="Y" if build == "GRCh37" else "chrY", ancestral=c["anc"], derived=c["der"])
            loci.append(dict(name=v["name"], snp=v["variantType"] == "SNP", coordinates=coords))
        all_nodes[str(idx)] = dict(haplogroup_id=idx, parent_id=parent_id, name=n["name"], is_root=is_root, loci=loci, children=[])
    for idx in range(len(doc)):
        node = all_nodes[str(idx)]
        if not node["is_root"]:
            all_nodes[str(node["parent_id"])]["children"].append(idx)
    return all_nodes


def build_tree(all_nodes, node_id):
    node = all_nodes.get(str(node_id))
    if node is None:
        return None
    children = [c for c in (build_tree(all_nodes, cid) for cid in node["children"]) if c is not None]
    if node["parent_id"] == 0:
        parent = None
    else:
        p = all_nodes.get(str(node["parent_id"]))
        if p is None:
            return None
        parent = p["name"]
    return dict(name=node["name"], parent=parent, loci=node["loci"], children=children)


def load_tree(text, provider):
    """provider: 'ftdna' | 'decodingus'.  -> (all_nodes, Haplogroup dict)"""
    if provider == "decodingus":
        all_nodes = parse_decodingus(text)
        roots = [n for n in all_nodes.values() if n["is_root"]]
        if not roots:
            raise TreeError("No node marked as root found in DecodingUs tree")
        root = roots[0]
    else:
        all_nodes = parse_ftdna(text)
        roots = [n for n in all_nodes.values() if n["parent_id"] == 0]
        if not roots:
            raise TreeError("No root node found in FTDNA tree")
        if len(roots) > 1:
            raise TreeError("Multiple root nodes found in FTDNA tree")
        root = roots[0]
    tree = build_tree(all_nodes, root["haplogroup_id"])
    if tree is None:
        raise TreeError("Failed to build tree")
    return all_nodes, tree


def count_nodes(h):
    return 1 + sum(count_nodes(c) for c in h["children"])


def collect_snps(h, positions, build_id):
    """positions: {vcf_pos: [(haplogroup name, locus)]}"""
    for locus in h["loci"]:
        coord = locus["coordinates"].get(build_id)
        if coord is not None and locus["snp"]:
            positions.setdefault(coord["position"], []).append((h["name"], locus))
    for c in h["children"]:
        collect_snps(c, positions, build_id)


def sites_and_relevance(tree, build_id, ref_name):
    positions = {}
    collect_snps(tree, positions, build_id)
    sites = sorted(positions)
    rel = [any((l["coordinates"].get(build_id) or {}).get("chromosome") == ref_name for _, l in positions[p]) for p in sites]
    return sites, rel


def call_sites(sites, relevant, hist, min_depth):
    """hist: per site 16 counts by 4-bit base code.  -> {pos: (base, total, freq)}"""
    calls = {}
    for i, pos in enumerate(sites):
        if not relevant[i]:
            continue
        h = [int(x) for x in hist[i]]
        total = sum(h)
        if total == 0 or total < min_depth:
            continue
        best = max(h)
        if h.count(best) > 1 and best / total >= 0.7:
            raise AssertionError("a >= 0.7 majority cannot tie")
        freq = best / total
        if freq >= 0.7:
            calls[pos] = (CODE[h.index(best)], total, freq)
    return calls


def calculate_haplogroup_score(h, snp_calls, scores, parent_info, depth, build_id):
    cur = dict(matches=0, ancestral_matches=0, no_calls=0, total_snps=0, score=0.0)
    cumulative = set(parent_info[1]) if parent_info is not None else set()
    defining = [l for l in h["loci"] if l["snp"] and build_id in l["coordinates"]]
    for l in defining:
        cumulative.add(l["coordinates"][build_id]["position"])
    derived = ancestral = no_calls = low_q = 0
    for l in defining:
        coord = l["coordinates"][build_id]
        call = snp_calls.get(coord["position"])
        if call is not None:
            base, dep, freq = call
            if dep >= 4:
                if not coord["derived"] or not coord["ancestral"]:
                    raise TreeError("empty allele")          # `.chars().next().unwrap()` panics
                d, a = coord["derived"][0], coord["ancestral"][0]
                if base == d:
                    if freq >= 0.7: derived += 1
                    elif freq >= 0.5: derived += 1
                    else: low_q += 1
                elif base == a:
                    if freq >= 0.7: ancestral += 1
                    else: low_q += 1
                elif freq >= 0.7:
                    derived += 1
                else:
                    low_q += 1
            else:
                no_calls += 1
        else:
            no_calls += 1
    if derived + ancestral + low_q > 0:
        if ancestral == 0:
            branch = 3.08 if derived >= 1 else 1.0
        else:
            d, a = derived, ancestral
            if d >= 3 and a <= d // 2: branch = 2.8
            elif d >= 2 and a <= d: branch = 2.5
            elif d >= 2: branch = 2.0
            elif d == 1 and a <= 2: branch = 1.5
            elif a > d * 3: branch = 0.0
            else: branch = 1.0
        cur["score"] = branch * (1.1 if low_q == 0 else 0.9)
    cur["matches"] += derived
    cur["ancestral_matches"] += ancestral
    cur["no_calls"] += no_calls
    cur["total_snps"] += len(defining)
    if ancestral > derived * 10:
        scores.append(dict(name=h["name"], score=0.0, matching_snps=derived, mismatching_snps=low_q, ancestral_matches=ancestral,
                           no_calls=no_calls, total_snps=len(defining), cumulative_snps=len(cumulative), depth=depth))
        return cur, cumulative
    for child in h["children"]:
        cs, cc = calculate_haplogroup_score(child, snp_calls, scores, (dict(cur), set(cumulative)), depth + 1, build_id)
        scores.append(dict(name=child["name"], score=cs["score"], matching_snps=cs["matches"], mismatching_snps=low_q,
                           ancestral_matches=cs["ancestral_matches"], no_calls=cs["no_calls"], total_snps=len(defining),
                           cumulative_snps=len(cc), depth=depth))
    return cur, cumulative


def find_path_to_root(h, target):
    if h["name"] == target:
        return [h["name"]]
    for c in h["children"]:
        p = find_path_to_root(c, target)
        if p is not None:
            p.append(h["name"])
            return p
    return None


def collect_scored_paths(scores, tree):
    unique = {}
    for r in scores:
        if r["name"] not in unique:
            unique[r["name"]] = r
        elif r["score"] > unique[r["name"]]["score"]:
            unique[r["name"]] = r
    remaining = [r for r in unique.values() if r["score"] > 0.0 and r["ancestral_matches"] <= r["matching_snps"] * 3 and r["matching_snps"] > 0]
    key = lambda r: (-r["cumulative_snps"], -r["score"], r["name"].encode())
    remaining.sort(key=key)
    ordered = []
    if remaining:
        path = find_path_to_root(tree, remaining[0]["name"])
        if path is not None:
            for name in path:
                for i, r in enumerate(remaining):
                    if r["name"] == name:
                        ordered.append(remaining.pop(i))
                        break
    remaining.sort(key=key)
    return ordered + remaining


def find_haplogroup(h, name):
    if h["name"] == name:
        return h
    for c in h["children"]:
        f = find_haplogroup(c, name)
        if f is not None:
            return f
    return None


def report_text(tree, snp_calls, build_id, show_snps):
    scores = []
    calculate_haplogroup_score(tree, snp_calls, scores, None, 0, build_id)
    rows = collect_scored_paths(scores, tree)
    out = ["Haplogroup\tScore\tMatching_SNPs\tMismatching_SNPs\tAncestral_Matches\tNo_Calls\tTotal_SNPs\tCumulative_SNPs\tDepth"
           + ("\tMatching_SNP_Details\tMismatching_SNP_Details\tNo_Call_Details" if show_snps else "")]
    for r in rows:
        line = "%s\t%.4f\t%d\t%d\t%d\t%d\t%d\t%d\t%d" % (r["name"], r["score"], r["matching_snps"], r["mismatching_snps"], r["ancestral_matches"],
                                                        r["no_calls"], r["total_snps"], r["cumulative_snps"], r["depth"])
        if show_snps:
            m, mm, nc = [], [], []
            node = find_haplogroup(tree, r["name"])
            if node is not None:
                for l in node["loci"]:
                    coord = l["coordinates"].get(build_id)
                    if coord is None:
                        continue
                    item = "%s:%d" % (l["name"], coord["position"])
                    call = snp_calls.get(coord["position"])
                    if call is None:
                        nc.append(item)
                    elif call[0] == coord["derived"][0]:
                        m.append(item)
                    else:
                        mm.append(item)
            line += "\t" + ";".join(m) + "\t" + ";".join(mm) + "\t" + ";".join(nc)
        out.append(line)
    return "\n".join(out) + "\n", rows


def validate_reference(header_text, ref_names, tree_type):
    """tree_type: 'Y' | 'MT'.  -> (build_id, chromosome)"""
    genome = report_oracle.reference_build(header_text)
    if genome == "Unknown":
        raise TreeError("Could not determine reference genome from BAM header")
    if tree_type == "MT":
        cand = ["chrM", "MT", "M"]
    else:
        cand = {"GRCh38": ["chrY", "Y", "NC_000024.10", "CM000686.2"], "GRCh37": ["Y", "chrY"],
                "T2T-CHM13v2.0": ["Y", "chrY", "CP086569.2", "NC_060948.1"]}[genome]
    for c in cand:
        if c in ref_names:
            return ("rCRS" if tree_type == "MT" else genome), c
    raise TreeError("No valid sequence found in BAM. Tried: " + ", ".join(cand))
