"""The CPython glue modules pyniNVStrings / pyniNVCategory / pyniNVText (custrings_amd/host/pyni_*.cpp):
the reference's module names, n_* function names and positional conventions (python/cpp/pystrings.cpp:
212-250, 1619-1642, 1902-1931, 2588-2666, 3860-3973), over libNVStrings.so & co.
CPU: they import, export the names the reference's Python classes call for the covered methods, and
raise ValueError (not crash) without a GPU.  GPU: the reference's test vectors through the n_* calls."""
import os
import re

import pytest

import cpulibs

ROOT = cpulibs.ROOT


def _mods():
    import subprocess

    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "custrings_amd", "host")], check=True)
    import pyniNVCategory
    import pyniNVStrings
    import pyniNVText

    return pyniNVStrings, pyniNVCategory, pyniNVText


COVERED = {
    "nvstrings.py": ("pyniNVStrings", "to_device from_offsets to_host to_offsets size len byte_count null_count set_null_bitmask copy split rsplit "
                     "split_record rsplit_record partition rpartition replace replace_multi replace_with_backrefs lstrip strip rstrip lower upper "
                     "find rfind find_from find_multiple compare match_strings startswith endswith contains match count findall findall_record extract extract_record sort order gather sublist scatter scalar_scatter "
                     "remove_strings add_strings cat join"),
    "nvcategory.py": ("pyniNVCategory", "to_device from_offsets from_strings from_strings_list size keys_size keys indexes_for_key value_for_index value "
                      "values values_cpointer add_strings remove_strings to_strings gather_strings gather gather_and_remap merge_category "
                      "merge_and_remap add_keys remove_keys remove_unused_keys set_keys"),
    "nvtext.py": ("pyniNVText", "tokenize unique_tokens token_count tokens_counts replace_tokens normalize_spaces ngrams"),
}


def test_pyni_modules_import_and_fail_loudly_without_gpu():
    s, c, t = _mods()
    import torch  # noqa: F401

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ValueError):
        s.n_createFromHostStrings(["a"])  # std::runtime_error -> ValueError, as the reference's glue does
    with pytest.raises(ValueError):
        s.n_createFromHostStrings(12)


@pytest.mark.skipif(not os.path.isdir("/root/reference/python"), reason="the reference tree is not on this box")
@pytest.mark.parametrize("pyfile", sorted(COVERED))
def test_pyni_exports_what_the_reference_python_calls(pyfile):
    """Every pyniNV*.n_* function that the reference's Python module calls inside the methods this
    build covers exists in the rebuilt glue module (so those methods of the reference's own Python
    classes run on it unchanged)."""
    mods = dict(zip(("pyniNVStrings", "pyniNVCategory", "pyniNVText"), _mods()))
    modname, methods = COVERED[pyfile]
    mod = mods[modname]
    src = open(os.path.join("/root/reference/python", pyfile)).read()
    # split the reference module into top-level functions and class methods by their def lines
    blocks = re.split(r"\n(?=(?:    )?def )", src)
    missing = []
    for blk in blocks:
        m = re.match(r"(?:    )?def (\w+)\(", blk)
        if not m or m.group(1) not in methods.split():
            continue
        for fn in re.findall(modname + r"\.(n_\w+)\(", blk):
            if not hasattr(mod, fn):
                missing.append((m.group(1), fn))
    assert not missing, missing


@pytest.mark.gpu
def test_gpu_pyni_strings_path():
    s, c, t = _mods()
    h = s.n_createFromHostStrings(["Héllo thesé", None, "are some", "tést String", ""])
    assert s.n_size(h) == 5
    assert s.n_createHostStrings(h) == ["Héllo thesé", None, "are some", "tést String", ""]
    cols = s.n_split(h, "s", None)  # cpp/tests/test_split.cpp:36-45
    assert [s.n_createHostStrings(x) for x in cols] == [["Héllo the", None, "are ", "té", ""], ["é", None, "ome", "t String", None]]
    recs = s.n_split_record(h, None, -1)
    assert [None if r is None else s.n_createHostStrings(r) for r in recs] == [["Héllo", "thesé"], None, ["are", "some"], ["tést", "String"], [""]]
    r = s.n_replace(h, "s", "Z", -1, False)
    assert s.n_createHostStrings(r) == ["Héllo theZé", None, "are Zome", "téZt String", ""]
    r2 = s.n_replace(h, "[st]+", "_", 1, True)
    assert s.n_createHostStrings(r2) == ["Héllo _hesé", None, "are _ome", "_ést String", ""]
    with pytest.raises(ValueError):
        s.n_replace(h, "", "x", -1, True)  # std::invalid_argument -> ValueError (pystrings.cpp:1912-1931)
    assert s.n_contains(h, "é", False, 0) == [True, None, False, True, False]
    assert s.n_contains(h, "^a", True, 0) == [False, None, True, False, False]
    assert s.n_find(h, "é", 0, None, 0) == [1, None, -1, 1, -1]
    assert s.n_len(h, 0) == [11, None, 8, 11, 0]
    # the rest of the find family (python/tests/test_compare.py:10-102)
    f = s.n_createFromHostStrings(["hello", "there", "world", "accéntéd", None, ""])
    assert s.n_compare(f, "there", 0) == [-12, 0, 3, -19, None, -1]
    assert s.n_rfind(f, "d", 0, None, 0) == [-1, -1, 4, 7, None, -1]
    assert s.n_find_from(f, "r", 0, 0, 0) == [-1, 3, 2, -1, None, -1]
    assert s.n_find_multiple(f, ["e", "o", "d"], 0) == [[1, 4, -1], [2, -1, -1], [-1, 1, 4], [-1, -1, 7], [None, None, None], [-1, -1, -1]]
    assert s.n_startswith(f, "he", 0) == [True, False, False, False, None, False]
    assert s.n_endswith(f, "d", 0) == [False, False, True, True, None, False]
    import nvstrings as _nvs

    other = _nvs.to_device(["hello", "there", "world", "accéntéd", None, ""])  # (an nvstrings object: the glue reads its m_cptr)
    assert s.n_match_strings(s.n_createFromHostStrings(["hello", "here", None, "accéntéd", None, ""]), other, 0) == [True, False, False, True, True, True]
    assert s.n_match_strings(f, ["hello", "there", "world", "accéntéd", None, ""], 0) == [True] * 6
    with pytest.raises(ValueError):
        s.n_match_strings(f, ["x"], 0)
    assert s.n_createHostStrings(s.n_upper(h))[0] == "HÉLLO THESÉ"
    assert s.n_createHostStrings(s.n_strip(s.n_createFromHostStrings(["  a  ", None]), None)) == ["a", None]
    assert s.n_createHostStrings(s.n_gather(h, [3, 0], 0)) == ["tést String", "Héllo thesé"]
    assert s.n_createHostStrings(s.n_gather(h, [True, False, False, False, True], 0)) == ["Héllo thesé", ""]
    # typed buffers are read by their item size: numpy's default int64 indexes, int16, and a bool array as a mask
    import numpy as np
    assert s.n_createHostStrings(s.n_gather(h, np.array([3, 0]), 0)) == ["tést String", "Héllo thesé"]
    assert s.n_createHostStrings(s.n_gather(h, np.array([3, 0], dtype=np.int16), 0)) == ["tést String", "Héllo thesé"]
    assert s.n_createHostStrings(s.n_gather(h, np.array([3, 0], dtype=np.int32), 0)) == ["tést String", "Héllo thesé"]
    assert s.n_createHostStrings(s.n_gather(h, np.array([True, False, False, False, True]), 0)) == ["Héllo thesé", ""]
    with pytest.raises(ValueError):
        s.n_gather(h, np.array([1.5, 2.0]), 0)
    with pytest.raises(ValueError):
        s.n_gather(h, [9], 0)  # std::out_of_range
    assert s.n_order(h, 2, True, True, 0) == [1, 4, 0, 2, 3]
    assert s.n_createHostStrings(s.n_cat(h, None, ":", "_")) == ["Héllo thesé:_:are some:tést String:"]
    m = s.n_replace_multi(s.n_createFromHostStrings(["hello there, good friend!", None]), [",", "!", "e"], s.n_createFromHostStrings(["_"]), True)
    assert s.n_createHostStrings(m) == ["h_llo th_r__ good fri_nd_", None]
    for x in cols + [r, r2, m, h]:
        s.n_destroyStrings(x)
    # category + text through their glue modules (python/tests/test_category.py:33-45, test_text.py:40-57)
    import nvstrings  # the Python objects the reference's glue reads m_cptr from

    e = nvstrings.to_device(["eee", "aaa", "eee", "ddd", "ccc", "ccc", "ccc", "eee", "aaa"])
    cat = c.n_createCategoryFromNVStrings(e)
    assert c.n_keys_size(cat) == 4 and c.n_size(cat) == 9
    assert s.n_createHostStrings(c.n_get_keys(cat)) == ["aaa", "ccc", "ddd", "eee"]
    assert c.n_get_values(cat, 0) == [3, 0, 3, 2, 1, 1, 1, 3, 0]
    assert c.n_get_value_for_string(cat, "ccc") == 1 and c.n_get_indexes_for_key(cat, "ccc", 0) == [4, 5, 6]
    g = c.n_gather_and_remap(cat, [1, 3, 1], 0)
    assert s.n_createHostStrings(c.n_get_keys(g)) == ["ccc", "eee"] and c.n_get_values(g, 0) == [0, 1, 0]
    with pytest.raises(ValueError):
        c.n_gather_strings(cat, [0, 4], 0)
    c.n_destroyCategory(g)
    c.n_destroyCategory(cat)
    tx = nvstrings.to_device(["the quick brown fox jumped over the lazy brown dog", "the sable siamésé cat jumped under the brown sofa", None, ""])
    assert t.n_token_count(tx, " ", 0) == [10, 9, 0, 0] and t.n_token_count(tx, "o", 0) == [6, 3, 0, 0]
    tk = t.n_tokenize(nvstrings.to_device(["a b", None, "c"]), None)
    assert s.n_createHostStrings(tk) == ["a", "b", "c"]
    s.n_destroyStrings(tk)
