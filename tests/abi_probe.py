"""C probe for the ABI of the drop-in boundary: prints sizeof(T) and offsetof(T, field) for every struct a caller of minigraph.h /
mgpriv.h shares with the library.  Compiled once against the reference's headers (golden, tests/golden/make_abi_layout.py) and once
against include/minigraph_amd.h (tests/test_abi.py)."""
import os
import subprocess
import tempfile

STRUCTS = {  # type -> fields whose offsets are part of the ABI (bit-fields cannot be probed with offsetof: the neighbours pin them)
    "mg128_t": ["x", "y"],
    "mg128_v": ["n", "m", "a"],
    "mg_idxopt_t": ["w", "k", "bucket_bits"],
    "mg_mapopt_t": ["flag", "mini_batch_size", "seed", "max_qlen", "pe_ori", "occ_max1", "occ_max1_cap", "occ_max1_frac", "bw", "bw_long",
                    "rmq_size_cap", "rmq_rescue_size", "rmq_rescue_ratio", "max_gap_pre", "max_gap", "max_gap_ref", "max_frag_len", "div",
                    "chn_pen_gap", "chn_pen_skip", "max_lc_skip", "max_lc_iter", "max_gc_skip", "min_lc_cnt", "min_lc_score", "min_gc_cnt",
                    "min_gc_score", "gdp_max_ed", "lc_max_trim", "lc_max_occ", "mask_level", "sub_diff", "best_n", "pri_ratio", "ref_bonus",
                    "cap_kalloc", "min_cov_mapq", "min_cov_blen"],
    "mg_ggopt_t": ["flag", "algo", "min_mapq", "min_map_len", "min_depth_len", "min_var_len", "match_pen", "ggs_shrink_pen",
                   "ggs_min_end_cnt", "ggs_min_end_frac", "ggs_max_iden", "ggs_min_inv_iden"],
    "mg_idx_t": ["g", "es", "b", "w", "k", "flag", "n_seg", "B"],
    "mg_lchain_t": ["off", "v", "rs", "re", "qs", "qe", "score", "dist_pre", "hash_pre"],
    "mg_llchain_t": ["off", "cnt", "v", "score", "ed"],
    "mg_cigar_t": ["n_cigar", "mlen", "blen", "aplen", "ss", "ee", "cigar"],
    "mg_ds_t": ["len", "n_off", "off", "ds"],
    "mg_gchain_t": ["id", "parent", "off", "cnt", "n_anchor", "score", "qs", "qe", "plen", "ps", "pe", "blen", "mlen", "div", "hash", "subsc",
                    "n_sub", "p", "ds"],
    "mg_gchains_t": ["km", "n_gc", "n_lc", "n_a", "rep_len", "gc", "lc", "a"],
    "gfa_arc_t": ["v_lv", "w", "rank", "ov", "ow"],
    "gfa_aux_t": ["m_aux", "l_aux", "aux"],
    "gfa_seg_t": ["len", "snid", "soff", "rank", "name", "seq", "utg", "aux"],
    "gfa_sseq_t": ["name", "min", "max", "rank"],
    "gfa_t": ["m_seg", "n_seg", "max_rank", "seg", "h_names", "m_sseq", "n_sseq", "sseq", "h_snames", "m_arc", "n_arc", "arc", "link_aux", "idx"],
    "gfa_edseq_t": ["seq", "len"],
    "kstring_t": ["l", "m", "s"],
}


def probe_source(includes):
    lines = ["#include <stdio.h>", "#include <stddef.h>"] + list(includes) + ["int main(void) {"]
    for t, fields in STRUCTS.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (t, t))
        for f in fields:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (t, f, t, f))
    lines += ["return 0; }"]
    return "\n".join(lines) + "\n"


def run_probe(cflags, includes):
    d = tempfile.mkdtemp(prefix="abi_probe_")
    src, exe = os.path.join(d, "probe.c"), os.path.join(d, "probe")
    open(src, "w").write(probe_source(includes))
    subprocess.check_call(["gcc", "-std=gnu99", "-w"] + list(cflags) + [src, "-o", exe])
    out = subprocess.check_output([exe]).decode()
    return {k: int(v) for k, v in (l.split() for l in out.splitlines())}
