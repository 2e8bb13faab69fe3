"""Generates tests/golden/ref_spellcheck.json by IMPORTING THE REFERENCE's utils/spell_check_utils.py and infer_effocr.py
(/root/reference, this container only; the absent packages — symspellpy among them — are import-only stubs that raise when used,
see make_ref_golden.py) and recording inputs / outputs of

  create_homoglyph_dict / create_common_abbrev   utils/spell_check_utils.py:8-57      (the tables themselves)
  visual_spell_checker                           utils/spell_check_utils.py:155-245   over a SYNTHETIC word-frequency dictionary
                                                 (the reference's own comes from symspellpy's packaged file, absent here)
  depunctuate / is_number / is_word / is_initial / all_caps / majority_normalize      :79-152
  EffOCR.en_postprocess with spell_check=True    infer_effocr.py:370-410              (WORDDICT / SIMDICT / ABBREVSET are module globals there)

Only recorded DATA is committed.  Run:  python tests/golden/make_ref_spellcheck.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_ref_golden as G                                   # the stub finder + import_reference

WORDS = {"hello": 900, "world": 800, "the": 5000, "oil": 120, "coil": 15, "boil": 30, "ill": 60, "hill": 75, "lily": 12, "july": 200,
         "you": 4000, "van": 90, "yan": 1, "pat": 40, "bat": 55, "had": 700, "nad": 2, "cat": 300, "oat": 9, "fat": 80, "tat": 3,
         "at": 2500, "as": 2600, "zoo": 25, "soo": 1, "dog": 350, "god": 340, "is": 3000, "it": 2900, "lt": 1, "all": 2200, "ail": 4,
         "hall": 110, "ha": 30, "a": 6000, "i": 5500, "gas": 66, "total": 77, "fotal": 1, "ten": 210, "hen": 21, "quit": 50, "duit": 1}
LINES = [
    "he11o wor1d", "Hel1o W0rld", "the 0il is b0il", "c0i1 0f 0il", "heHo there", "HeHo", "1ily in ju1y", "y0u van yan", "YOU VAN", "hcd a bat",
    "Mr. Jones vs. Dr. Smith", "mr. j0nes", "M. Dupont", "1. item", "l. item", "J. K. Row1ing", "3,500 d0gs", "l23 and 4S6", "12O0 is 1200",
    "a-b c/d e'f", 'say "he11o" now', "it's 9as", "tota1 f0tal", "t0tal", "|t is", "i11 hi11 ha11", "2t @t", "ten hen", "TEN HEN", "q-uit",
    "", " ", "-", "'", '"', "x", "0", "O", "1", "l", "I", "z00", "zoo soo", "d09", "9od", "a11", "ai1", "i.e. e.g.", "l.e.", "0z.", "lnc.",
    "hello, world.", "he11o, wor1d!", "(he11o)", "0ne tw0 thr33", "h3llo", "he110", "1234x", "x1234", "1a1", "a1a", "1l1", "lOl", "l0l",
]


def main():
    multi, single, EffLocalizer, du = G.import_reference()
    import utils.spell_check_utils as U                       # imported by infer_effocr already (symspellpy stubbed)
    sim, abbr = U.create_homoglyph_dict(), U.create_common_abbrev()
    worddict = dict(WORDS)
    for a in abbr:                                            # create_worddict's last step (:19-23), on the synthetic dictionary
        worddict.pop(U.depunctuate(a), None)
    out = {"generated_by": "tests/golden/make_ref_spellcheck.py (imports /root/reference; stubs armed while recording)",
           "homoglyphs": sim, "abbrevs": sorted(abbr), "words": worddict, "checker": [], "helpers": [], "majority": [], "postprocess": []}
    for line in LINES:
        for beam, mn in ((1000, True), (1000, False), (3, True)):
            case = {"line": line, "beam": beam, "majority_norm": mn}
            try:
                case["out"] = U.visual_spell_checker(line, worddict, sim, abbr, beam=beam, majority_norm=mn)
            except Exception as e:
                case["raises"] = type(e).__name__
            out["checker"].append(case)
    probes = ["", "a", "A", "AB", "Ab", "J.", "j.", "JK", "1.", "12", "1,200", "$3.50", "3a", "mr", "Mr.", "hello", "HELLO", "he-llo", "x.", "..", "Q."]
    for s in probes:
        out["helpers"].append({"s": s, "depunctuate": U.depunctuate(s), "is_number": U.is_number(s), "is_word": U.is_word(s, worddict),
                               "is_initial": U.is_initial(s), "all_caps": U.all_caps(s), "is_abbrev": U.is_abbrev(s, abbr)})
    for s in ["he11o", "1a1", "a1a", "12O0", "l23", "4S6", "abc", "123", "a1", "1a", "h3llo", "x9x9x", "2t", "@t", "1|1", "1z1", "1b1", "o0o0"]:
        case = {"s": s}
        try:
            case["out"] = U.majority_normalize(s, sim)
        except Exception as e:
            case["raises"] = type(e).__name__
        out["majority"].append(case)
    # EffOCR.en_postprocess with the spell checker switched on (the index lists are computed BEFORE the line is corrected)
    single.WORDDICT, single.SIMDICT, single.ABBREVSET = worddict, sim, abbr
    rng = np.random.default_rng(7)
    texts = ["he11oworld", "theoi1isb0il", "Hel1oW0rld", "y0uvanyan", "t0ta1ten", "i11hi11", "heHothere", "aenr-aenr", "d09andcat"]
    for c, t in enumerate(texts * 2):
        ns = types.SimpleNamespace(LARGE_NUM=1_000_000, anchor_multiplier=4, anchor_margin=[None, 0.15][c % 2], spell_check=True)
        n = len(t)
        wl = sorted(set(int(v) for v in rng.integers(0, n, 3)) | {0})
        base = float(rng.uniform(12, 24))
        heights = [float(base * (1.0 if ch in "aenrwuosvcxz-." else 1.5) * rng.uniform(0.95, 1.05)) for ch in t]
        bottoms = [float(40.0 + rng.uniform(-1, 1)) for _ in t]
        case = {"line": t, "word_end_idx": wl, "heights": heights, "bottoms": bottoms, "anchor_margin": ns.anchor_margin}
        try:
            case["out"] = single.EffOCR.en_postprocess(ns, t, wl, heights, bottoms)
        except Exception as e:
            case["raises"] = type(e).__name__
        out["postprocess"].append(case)
    with open(os.path.join(HERE, "ref_spellcheck.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=0, sort_keys=True)
    print(f"recorded {len(out['checker'])} checker cases, {len(out['helpers'])} helper probes, {len(out['majority'])} majority cases, "
          f"{len(out['postprocess'])} en_postprocess cases")
    for c in out["checker"][:60:3]:
        print(repr(c["line"]), "->", repr(c.get("out", c.get("raises"))))
    for c in out["postprocess"]:
        print(repr(c["line"]), c["anchor_margin"], "->", repr(c.get("out", c.get("raises"))))


if __name__ == "__main__":
    main()
