#!/usr/bin/env python3
"""Container-only: pin the reference's STATIC DATA by text.

Parses the initialisers and numeric literals of /root/reference/src/hfdl.c (and the three #defines of src/hfdl.h) into
tests/golden/hfdl_constants.json -- numbers only, no source text is stored.  The oracle's tables, the product's tables
(csrc/demod_tables.h / demod_logic.h through tests/hostsim) and the tables resident on the device are compared with this file
(tests/test_constants_cpu.py, tests/test_gpu_constants.py): a typo common to the hand-typed copies is no longer invisible.

What is read (reference file:line as of the surveyed tree):
  src/hfdl.h:6-8            SPS, HFDL_SYMBOL_RATE, HFDL_CHANNEL_TRANSITION_BW_HZ
  src/hfdl.c:29-46          PREKEY_LEN ... HFDL_SSB_CARRIER_OFFSET_HZ (the derived lengths are evaluated)
  src/hfdl.c:48-70          the sampler / framer / modulation enumerators
  src/hfdl.c:81-138         hfdl_frame_params[8]
  src/hfdl.c:144-154        SYMSYNC_PFB_CNT, HFDL_MF_SYMBOL_DELAY, HFDL_MF_TAPS_CNT, hfdl_matched_filter[19]
  src/hfdl.c:157-160        T_seq[2][15]
  src/hfdl.c:252-253        Costas loop gains
  src/hfdl.c:325-341        descrambler LFSR parameters, both liquid API branches
  src/hfdl.c:419-447        A_octets[16], M1_bits[127], M_shifts[8]
  src/hfdl.c:470-505        constructor arguments of the liquid-dsp objects
  src/hfdl.c:613,663-665    13-frame time-out, time-stamp correction
  src/hfdl.c:699-712        noise-floor estimator, carrier run-away threshold

Run:  python tests/golden/make_constants.py     (needs /root/reference; the GPU box never runs it)
"""
import json
import os
import re
import sys

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def number(tok):
    tok = tok.strip()
    if re.fullmatch(r"0b[01]+", tok):
        return int(tok[2:], 2)
    if re.fullmatch(r"0x[0-9a-fA-F]+[uUlL]*", tok):
        return int(re.sub(r"[uUlL]+$", "", tok), 16)
    if re.fullmatch(r"[+-]?\d+[uUlL]*", tok):
        return int(re.sub(r"[uUlL]+$", "", tok))
    return float(re.sub(r"[fF]$", "", tok))


def defines(text):
    out = {}
    for m in re.finditer(r"^[ \t]*#define[ \t]+(\w+)[ \t]+(.+?)[ \t]*$", text, flags=re.M):
        out[m.group(1)] = m.group(2).strip()
    return out


def eval_define(name, table, seen=()):
    """integer / float value of a #define whose body is literals, other #defines, + - * and parentheses"""
    if name in seen:
        raise ValueError("recursive define " + name)
    body = table[name]

    def sub(m):
        w = m.group(0)
        if w in table:
            return repr(eval_define(w, table, seen + (name,)))
        return w
    expr = re.sub(r"[A-Za-z_]\w*", sub, body)
    expr = re.sub(r"(\d)[fFuUlL]+\b", r"\1", expr)
    if not re.fullmatch(r"[0-9eE+\-*/(). x]+", expr):
        raise ValueError("define %s: cannot evaluate %r" % (name, body))
    return eval(expr, {"__builtins__": {}})


def brace_block(text, start):
    """text of the balanced {...} that begins at or after `start`"""
    i = text.index("{", start)
    depth, j = 0, i
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j]
        j += 1


def numbers_in(block):
    return [number(t) for t in re.findall(r"0b[01]+|0x[0-9a-fA-F]+[uUlL]*|[+-]?(?:\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+(?:[eE][+-]?\d+)?)[fFuUlL]*", block)]


def call_args(text, func, nth=0):
    ms = list(re.finditer(re.escape(func) + r"\s*\(", text))
    m = ms[nth]
    depth, j = 1, m.end()
    while depth:
        depth += {"(": 1, ")": -1}.get(text[j], 0)
        j += 1
    return [a.strip() for a in text[m.end():j - 1].split(",")]


def main():
    if not os.path.isdir(REF):
        sys.exit("needs %s (this script runs in the build container only)" % REF)
    c = strip_comments(open(os.path.join(REF, "hfdl.c")).read())
    h = strip_comments(open(os.path.join(REF, "hfdl.h")).read())
    table = defines(h)
    table.update(defines(c))
    enums = {}
    for m in re.finditer(r"\b([A-Z][A-Z0-9_]+)\s*=\s*(\d+)\s*[,}\n]", c[:c.index("struct hfdl_params")]):
        enums[m.group(1)] = int(m.group(2))
    table.update({k: str(v) for k, v in enums.items()})
    want = ["SPS", "HFDL_SYMBOL_RATE", "HFDL_CHANNEL_TRANSITION_BW_HZ", "PREKEY_LEN", "A_LEN", "M1_LEN", "M2_LEN", "M_SHIFT_CNT", "T_LEN", "EQ_LEN",
            "DATA_FRAME_LEN", "DATA_FRAME_CNT_SINGLE_SLOT", "DATA_FRAME_CNT_DOUBLE_SLOT", "DATA_SYMBOLS_CNT_MAX", "PREAMBLE_LEN",
            "SINGLE_SLOT_FRAME_LEN", "CORR_THRESHOLD_A1", "CORR_THRESHOLD_A2", "CORR_THRESHOLD_M1", "MAX_SEARCH_RETRIES",
            "HFDL_SSB_CARRIER_OFFSET_HZ", "SYMSYNC_PFB_CNT", "HFDL_MF_SYMBOL_DELAY", "HFDL_MF_TAPS_CNT", "MOD_ARITY_MAX"]
    out = {"generated_by": "tests/golden/make_constants.py from /root/reference/src/hfdl.c, src/hfdl.h (numbers only)",
           "defines": {k: eval_define(k, table) for k in want}, "enums": enums}

    # hfdl_frame_params[8]: designated initialisers [i] = { .scheme = M_x, .data_segment_cnt = ..., .code_rate = n, .deinterleaver_push_column_shift = n }
    blk = brace_block(c, c.index("hfdl_frame_params[M_SHIFT_CNT]"))
    params = {}
    for m in re.finditer(r"\[(\d+)\]\s*=\s*\{(.*?)\}", blk, flags=re.S):
        f = dict((k, v.strip()) for k, v in re.findall(r"\.(\w+)\s*=\s*([^,}]+)", m.group(2)))
        val = lambda s: int(eval_define(s, table)) if s in table else int(number(s))
        params[int(m.group(1))] = [val(f["scheme"]), val(f["data_segment_cnt"]), val(f["code_rate"]), val(f["deinterleaver_push_column_shift"])]
    assert sorted(params) == list(range(8))
    out["frame_params"] = {"fields": ["bits_per_symbol", "data_segment_cnt", "code_rate", "deinterleaver_push_column_shift"], "modes": [params[i] for i in range(8)]}

    mf = brace_block(c, c.index("hfdl_matched_filter[HFDL_MF_TAPS_CNT]"))
    out["matched_filter"] = numbers_in(mf)
    assert len(out["matched_filter"]) == out["defines"]["HFDL_MF_TAPS_CNT"]

    tblk = brace_block(c, c.index("T_seq[2][T_LEN]"))
    rows = {}
    for m in re.finditer(r"\[(\d)\]\s*=\s*\{(.*?)\}", tblk, flags=re.S):
        rows[int(m.group(1))] = [float(x) for x in numbers_in(m.group(2))]
    out["T_seq"] = [rows[0], rows[1]]
    assert all(len(r) == out["defines"]["T_LEN"] for r in out["T_seq"])

    out["A_octets"] = numbers_in(brace_block(c, c.index("A_octets[]")))
    out["M1_bits"] = numbers_in(brace_block(c, c.index("M1_bits[M1_LEN]")))
    out["M_shifts"] = numbers_in(brace_block(c, c.index("M_shifts[M_SHIFT_CNT]")))
    assert len(out["A_octets"]) == 16 and len(out["M1_bits"]) == 127 and len(out["M_shifts"]) == 8

    # Costas loop: c->alpha = 0.1f; c->beta = 0.047f * c->alpha * c->alpha;
    alpha = number(re.search(r"c->alpha\s*=\s*([0-9.eE+-]+f?)\s*;", c).group(1))
    beta_k = number(re.search(r"c->beta\s*=\s*([0-9.eE+-]+f?)\s*\*\s*c->alpha\s*\*\s*c->alpha", c).group(1))
    runaway = number(re.search(r"fabsf\(c->loop->dphi\)\s*>\s*([0-9.eE+-]+f?)", c).group(1))
    out["costas"] = {"alpha": alpha, "beta_over_alpha_squared": beta_k, "runaway_dphi": runaway,
                     "limit": number(re.search(r"branchless_limit\(c->err,\s*([0-9.eE+-]+f?)\)", c).group(1))}

    # descrambler: both API branches
    d = c[c.index("hfdl_descrambler_create(void)"):]
    d = d[:d.index("return descrambler_create")]
    polys = [number(x) for x in re.findall(r"lfsr_genpoly\s*=\s*(0x[0-9a-fA-F]+u?)", d)]
    inits = [number(x) for x in re.findall(r"lfsr_init\s*=\s*(0x[0-9a-fA-F]+u?)", d)]
    out["descrambler"] = {"numbits": number(re.search(r"numbits\s*=\s*(\d+)", d).group(1)), "seq_len": number(re.search(r"seq_len\s*=\s*(\d+)", d).group(1)),
                          "liquid_before_1_6": {"genpoly": polys[0], "init": inits[0]}, "liquid_1_6_and_later": {"genpoly": polys[1], "init": inits[1]}}

    # constructor arguments of the liquid objects (hfdl_channel_create)
    cc = c[c.index("struct block *hfdl_channel_create("):]
    cc = cc[:cc.index("void hfdl_channel_destroy")]
    sym = lambda a: eval_define(a, table) if a in table else number(a)
    agc_bw = [number(call_args(cc, "agc_crcf_set_bandwidth", i)[1]) for i in range(len(re.findall(r"agc_crcf_set_bandwidth\s*\(", cc)))]
    ss = call_args(cc, "symsync_crcf_create_kaiser")
    out["constructors"] = {
        "msresamp_stopband_db": number(call_args(cc, "msresamp_crcf_create")[1]),
        "agc_bandwidth_calls": agc_bw, "agc_bandwidth": agc_bw[-1],
        "noise_floor_init": number(re.search(r"c->noise_floor\s*=\s*([0-9.eE+-]+f?)\s*;", cc).group(1)),
        "eqlms_lowpass": [sym(call_args(cc, "eqlms_cccf_create_lowpass")[0]), number(call_args(cc, "eqlms_cccf_create_lowpass")[1])],
        "eqlms_bw": number(call_args(cc, "eqlms_cccf_set_bw")[1]),
        "symsync_create_kaiser": [sym(ss[0]), sym(ss[1]), number(ss[2]), sym(ss[3])],
        "symsync_lf_bw": number(call_args(cc, "symsync_crcf_set_lf_bw")[1]),
        "symsync_output_rate": number(call_args(cc, "symsync_crcf_set_output_rate")[1]),
        "resamp_rate_numerator": [sym("HFDL_SYMBOL_RATE"), sym("SPS")],
    }

    # decoder thread: noise floor estimator, time-out, time-stamp correction
    t = c[c.index("static void *hfdl_decoder_thread"):]
    nf = re.search(r"c->noise_floor\s*=\s*([0-9.eE+-]+f?)\s*\*\s*c->noise_floor\s*\+\s*([0-9.eE+-]+f?)\s*\*\s*fminf\(.*?\)\s*\+\s*([0-9.eE+-]+f?)\s*;", t, flags=re.S)
    clk = re.search(r"\+\+noise_floor_sampling_clk\s*&\s*(0x[0-9A-Fa-f]+u?)\)\s*==\s*(0x[0-9A-Fa-f]+u?)", t)
    out["decoder_thread"] = {
        "noise_floor_keep": number(nf.group(1)), "noise_floor_take": number(nf.group(2)), "noise_floor_bias": number(nf.group(3)),
        "noise_floor_clk_mask": number(clk.group(1)), "noise_floor_clk_match": number(clk.group(2)),
        "max_frames_without_frame": number(re.search(r"max_symbols_without_frame\s*=\s*(\d+)\s*\*\s*SINGLE_SLOT_FRAME_LEN", t).group(1)),
        "ts_correction_symbols": eval_define("PREKEY_LEN", table) + 2 * eval_define("A_LEN", table),
    }
    path = os.path.join(HERE, "hfdl_constants.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
