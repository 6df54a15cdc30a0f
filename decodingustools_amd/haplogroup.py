"""Host-side mirror of the reference's haplogroup module around the site-list pileup
(include/dut_haplogroup.h; src/haplogroup/{mod,tree,caller,scoring,validation}.rs, src/vendor/*.rs)."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .callable_loci import EngineError

FTDNA, DECODINGUS = 0, 1          # cli::TreeProvider
YDNA, MTDNA = 0, 1                # utils::cache::TreeType


@dataclass
class SnpCall:
    position: int
    base: str
    depth: int
    freq: float


@dataclass
class HaplogroupResult:
    name: str
    score: float
    matching_snps: int
    mismatching_snps: int
    ancestral_matches: int
    no_calls: int
    total_snps: int
    cumulative_snps: int
    depth: int


def _calls_c(calls: Sequence[SnpCall]):
    arr = (_lib.dut_snp_call * max(len(calls), 1))()
    for i, c in enumerate(sorted(calls, key=lambda c: c.position)):
        arr[i].position = c.position; arr[i].depth = c.depth; arr[i].freq = c.freq; arr[i].base = c.base.encode()
    return arr


class HaplogroupTree:
    """provider.parse_tree + load_tree + build_tree on a JSON text of the provider's shape."""

    def __init__(self, json_text, provider: int = FTDNA, tree_type: int = YDNA, path: Optional[str] = None):
        self._lib = _lib.load()
        err = C.create_string_buffer(1024)
        if path is not None:
            self._h = self._lib.dut_tree_load(path.encode(), provider, tree_type, err, 1024)
        else:
            data = json_text.encode() if isinstance(json_text, str) else bytes(json_text)
            self._h = self._lib.dut_tree_parse(data, len(data), provider, tree_type, err, 1024)
        if not self._h:
            raise EngineError(-1, err.value.decode())

    @classmethod
    def load(cls, path: str, provider: int = FTDNA, tree_type: int = YDNA):
        return cls(None, provider, tree_type, path=path)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dut_tree_free(self._h)
            self._h = None

    __del__ = close

    @property
    def total_nodes(self) -> int:
        return self._lib.dut_tree_total_nodes(self._h)

    @property
    def built_nodes(self) -> int:
        return self._lib.dut_tree_built_nodes(self._h)

    @property
    def root_name(self) -> str:
        return self._lib.dut_tree_root_name(self._h).decode()

    def collect_sites(self, build_id: str, ref_name: str) -> Tuple[np.ndarray, np.ndarray]:
        """(positions ascending, relevant flags): tree::collect_snps + the chromosome test of caller.rs:96-103."""
        sp, rp, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        st = self._lib.dut_tree_collect_sites(self._h, build_id.encode(), ref_name.encode(), C.byref(sp), C.byref(rp), C.byref(n))
        if st != 0:
            raise EngineError(st, "collect_snps failed")
        try:
            sites = np.frombuffer(C.string_at(sp.value, 4 * n.value), dtype=np.uint32).copy() if n.value else np.zeros(0, np.uint32)
            rel = np.frombuffer(C.string_at(rp.value, n.value), dtype=np.uint8).copy() if n.value else np.zeros(0, np.uint8)
        finally:
            self._lib.dut_free(sp); self._lib.dut_free(rp)
        return sites, rel

    def score(self, calls: Sequence[SnpCall], build_id: str) -> List[HaplogroupResult]:
        """calculate_haplogroup_score + collect_scored_paths."""
        arr = _calls_c(calls)
        rp, n = C.c_void_p(), C.c_size_t()
        err = C.create_string_buffer(512)
        st = self._lib.dut_tree_score(self._h, arr, len(calls), build_id.encode(), C.byref(rp), C.byref(n), err, 512)
        if st != 0:
            raise EngineError(st, err.value.decode())
        try:
            res = C.cast(rp, C.POINTER(_lib.dut_haplogroup_result))
            return [HaplogroupResult(res[i].name.decode(), res[i].score, res[i].matching_snps, res[i].mismatching_snps,
                                     res[i].ancestral_matches, res[i].no_calls, res[i].total_snps, res[i].cumulative_snps,
                                     res[i].depth) for i in range(n.value)]
        finally:
            self._lib.dut_free(rp)

    def write_report(self, path: str, calls: Sequence[SnpCall], build_id: str, show_snps: bool = False):
        """score + the TSV of analyze_haplogroup (mod.rs:92-138)."""
        arr = _calls_c(calls)
        rp, n = C.c_void_p(), C.c_size_t()
        err = C.create_string_buffer(512)
        st = self._lib.dut_tree_score(self._h, arr, len(calls), build_id.encode(), C.byref(rp), C.byref(n), err, 512)
        if st != 0:
            raise EngineError(st, err.value.decode())
        try:
            st = self._lib.dut_write_haplogroup_report(path.encode(), self._h, rp, n.value, arr, len(calls), build_id.encode(),
                                                       1 if show_snps else 0, err, 512)
            if st != 0:
                raise EngineError(st, err.value.decode())
        finally:
            self._lib.dut_free(rp)


def call_sites(sites, hist, min_depth: int, relevant=None) -> List[SnpCall]:
    """The per-site call of process_region (caller.rs:132-149) from site_pileup's histograms."""
    lib = _lib.load()
    sites = np.ascontiguousarray(sites, np.uint32)
    hist = np.ascontiguousarray(hist, np.uint32)
    rel = np.ascontiguousarray(relevant, np.uint8) if relevant is not None else None
    cp, n = C.c_void_p(), C.c_size_t()
    st = lib.dut_call_sites(sites.ctypes.data, rel.ctypes.data if rel is not None else None, hist.ctypes.data, sites.shape[0],
                            min_depth, C.byref(cp), C.byref(n))
    if st != 0:
        raise EngineError(st, "calling failed")
    try:
        arr = C.cast(cp, C.POINTER(_lib.dut_snp_call))
        return [SnpCall(arr[i].position, arr[i].base.decode(), arr[i].depth, arr[i].freq) for i in range(n.value)]
    finally:
        lib.dut_free(cp)


def validate_reference(header_text: bytes, ref_names: Sequence[str], tree_type: int = YDNA) -> Tuple[str, str]:
    """(build_id, chromosome): validation::validate_reference + the build id choice of mod.rs:51-54."""
    lib = _lib.load()
    names = (C.c_char_p * max(len(ref_names), 1))(*[n.encode() for n in ref_names])
    b, c, e = C.create_string_buffer(64), C.create_string_buffer(256), C.create_string_buffer(512)
    st = lib.dut_validate_reference(header_text, len(header_text), names, len(ref_names), tree_type, b, 64, c, 256, e, 512)
    if st != 0:
        raise EngineError(st, e.value.decode())
    return b.value.decode(), c.value.decode()


def analyze_haplogroup(bam_file: str, reference_file: str, tree_json: str, output_file: str, min_depth: int = 10,
                       min_quality: int = 20, tree_type: int = YDNA, provider: int = FTDNA, show_snps: bool = False,
                       device_id: int = 0):
    """haplogroup::analyze_haplogroup (mod.rs:17-141) with the tree read from a local JSON file."""
    lib = _lib.load()
    err = C.create_string_buffer(1024)
    st = lib.dut_find_branch_files(bam_file.encode(), reference_file.encode(), tree_json.encode(), output_file.encode(),
                                   min_depth, min_quality, tree_type, provider, 1 if show_snps else 0, device_id, err, 1024)
    if st != 0:
        raise EngineError(st, err.value.decode())
