"""Homoglyph spell checker of the torch driver's English post-processing (the optional branch): `EffOCR(spell_check=True)`
runs `visual_spell_checker(line, WORDDICT, SIMDICT, ABBREVSET)` between computing the case-repair index lists and applying them
(infer_effocr.py:401-402; implementation utils/spell_check_utils.py:155-245, tables :8-57).  Host-side string code, no GPU work.

What it does, word by word (words = the pieces between the splitters space / slash / hyphen / double quote / apostrophe):
a piece that is neither a dictionary word, nor a number, nor all capitals is expanded into candidates by substituting, position by
position, every visually similar glyph ("homoglyph": 0/O, 1/l/i, v/y, ...); the most frequent dictionary word among the candidates wins,
else the first abbreviation, initial ("J.") or number found, else the piece stays.  Afterwards digits inside mostly-alphabetic pieces (and
letters inside mostly-numeric ones) are mapped to their first homoglyph of the other kind ("majority normalisation").

The reference takes its word frequencies from symspellpy's packaged `frequency_dictionary_en_82_765.txt` (create_worddict, :15-24); that
package is not a dependency here: `load_worddict(path)` reads any file of that format ("word count" per line) and
`create_worddict()` uses symspellpy's copy when the package is importable.  Behaviour (including the quirks listed below) is pinned to
fixtures recorded from the imported reference over a synthetic dictionary: tests/golden/make_ref_spellcheck.py, tests/test_ref_golden.py.

Quirks kept (they change outputs):
  * the splitter list is the pattern's text cut at "|", so the double quote is listed as backslash + quote: a lone '"' piece is NOT
    recognised as a splitter and goes through the word path (harmlessly);
  * substitutions index the candidate by the position in the ORIGINAL piece, so after the two-glyph substitution H -> "ll" later
    positions are off by one;
  * candidates are kept in a list cut to the last `beam` entries; a candidate already accepted as a word stays in the pool;
  * the frequency lookup uses the de-punctuated lower-case form, ties go to the earliest candidate.
"""
import re

_PUNCT_DROPPED = ',.?!$%&():;-"'
_DEPUNCT = str.maketrans("", "", _PUNCT_DROPPED)
# glyph -> substitutes, in the reference's order (utils/spell_check_utils.py:27-57); "H" -> the two-glyph string "ll"
_HOMOGLYPHS = ("0:O O:0,C o:0,c 1:l,i l:i,1 i:l,j,1 j:i I:l,1 |:l,i,1 v:y V:Y y:v q:d d:q p:b b:p h:n n:h c:o C:O f:t t:f 2:a @:a "
               ",:. -:. z:s 9:g H:ll")
_ABBREVS = "dr. est. i.e. jr. inc. ltd. mr. mrs. ms. oz. sr. vs. e.g."
DEFAULT_SPLITTER_PATTERN = r"( |/|-|\"|')"


def create_common_abbrev():
    return set(_ABBREVS.split())


def create_homoglyph_dict():
    table = {}
    for item in _HOMOGLYPHS.split(" "):
        glyph, alts = item[0], item[2:]
        table[glyph] = alts.split(",")
    return table


def depunctuate(s):
    return s.translate(_DEPUNCT)


def is_number(s):
    return depunctuate(s).isdigit()


def is_word(s, words):
    return depunctuate(s.lower()) in words


def is_initial(s):
    return len(s) == 2 and s[0].isalpha() and s[0].isupper() and s[1] == "."


def is_abbrev(s, abbrevs):
    return s.lower() in abbrevs


def all_caps(s):
    return all(ch.isupper() for ch in s)


def load_worddict(path, abbrevs=None):
    """A symspellpy-format frequency dictionary ("word count" per line) -> {word: count}, minus the de-punctuated common abbreviations
    (create_worddict, utils/spell_check_utils.py:15-24: "mr", "dr", ... must not pass as words)."""
    words = {}
    with open(path, encoding="utf-8") as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 2 and parts[1].isdigit():
                words[parts[0]] = int(parts[1])
    for a in (create_common_abbrev() if abbrevs is None else abbrevs):
        words.pop(depunctuate(a), None)
    return words


def create_worddict(path=None):
    if path is not None:
        return load_worddict(path)
    try:
        import pkg_resources
        return load_worddict(pkg_resources.resource_filename("symspellpy", "frequency_dictionary_en_82_765.txt"))
    except Exception as e:
        raise RuntimeError("spell_check needs a word-frequency dictionary: pass a symspellpy-format file (word count per line) to "
                           "create_worddict(path), or install symspellpy for its packaged frequency_dictionary_en_82_765.txt") from e


def _neighbour_ok(s, i, pred):
    return True if (i < 0 or i >= len(s)) else pred(s[i])


def majority_normalize(s, simdict):
    digits = sum(ch.isdigit() for ch in s)
    alphas = sum(ch.isalpha() for ch in s)
    if alphas == digits:
        return s
    out = []
    for i, ch in enumerate(s):
        if alphas > digits:
            hit = ch.isdigit() and _neighbour_ok(s, i - 1, str.isalpha) and _neighbour_ok(s, i + 1, str.isalpha) and ch in simdict
            out.append(simdict[ch][0] if hit else ch)
        else:
            hit = ch.isalpha() and _neighbour_ok(s, i - 1, str.isdigit) and _neighbour_ok(s, i + 1, str.isdigit) and ch in simdict
            # (the reference indexes the first DIGIT substitute and raises IndexError when the glyph has none — kept)
            out.append([x for x in simdict[ch] if x.isdigit()][0] if hit else ch)
    return "".join(out)


def _correct_piece(w, worddict, simdict, abbrevs, beam):
    pool = [w]
    words, numbers, initials, abbrs = [], [], [], []
    for pos, ch in enumerate(w):
        for alt in simdict.get(ch, ()):
            fresh = []
            for cand in pool:
                sub = cand[:pos] + alt + cand[pos + 1:]
                if is_word(sub, worddict):
                    words.append(sub)
                elif is_abbrev(sub, abbrevs):
                    abbrs.append(sub)
                elif is_number(sub):
                    numbers.append(sub)
                elif is_initial(sub):
                    initials.append(sub)
                fresh.append(sub)
            pool = (pool + fresh)[-beam:]
    if words:
        freqs = [worddict[depunctuate(x).lower()] for x in words]
        return words[freqs.index(max(freqs))]
    for found in (abbrs, initials, numbers):
        if found:
            return found[0]
    return w


def visual_spell_checker(textline, worddict, vsim_dict, abbrevset, beam=1000, splitter_pattern=DEFAULT_SPLITTER_PATTERN, majority_norm=True):
    """Drop-in for utils/spell_check_utils.py:155 (same arguments, same result)."""
    splitters = splitter_pattern[1:-1].split("|")
    pieces = []
    for w in re.split(splitter_pattern, textline):
        if len(w) > 0 and w not in splitters and not (is_word(w, worddict) or is_number(w) or all_caps(w)):
            pieces.append(_correct_piece(w, worddict, vsim_dict, abbrevset, beam))
        else:
            pieces.append(w)
    if majority_norm:
        pieces = [majority_normalize(w, vsim_dict) if (w not in splitters and not is_number(w)) else w for w in pieces]
    return "".join(pieces)
