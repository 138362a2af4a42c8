"""Formula specs shared by the golden generator (built with the reference's stl_d_lib) and the tests (built with
pstl_diffusion_policy_amd.stl_d_lib): nested lists -> formula objects of whichever library is passed in.
Leaves ["ap", i] read signal i of x (x: tensor (n_sig, n, T))."""

SPECS = {
    "alw_ap": ["alw", 0, 20, ["ap", 0]],
    "ev_win": ["ev", 2, 7, ["ap", 1]],
    "and_or_not": ["and", ["or", ["ap", 0], ["not", ["ap", 1]]], ["ap", 2]],
    "imply": ["alw", 0, 12, ["imply", ["ap", 0], ["ev", 0, 5, ["ap", 1]]]],
    "listand_path": ["listand", [["alw", 0, 20, ["ap", 0]], ["alw", 0, 20, ["ap", 1]],
                                 ["ev", 0, 10, ["alw", 0, 20, ["and", ["ap", 2], ["ap", 3]]]],
                                 ["ev", 0, 10, ["alw", 0, 20, ["ap", 0]]], ["alw", 0, 20, ["ap", 3]]]],
    "once": ["alw", 3, 20, ["once", -3, 0, ["ap", 0]]],
    "once_incl": ["once", -4, -1, ["ap", 2]],
    "untimed_until": ["uu", ["ap", 0], ["ap", 1]],
    "until_0": ["until", 0, 6, ["ap", 2], ["ap", 3]],
    "until_3_9": ["until", 3, 9, ["ap", 0], ["and", ["ap", 1], ["ap", 3]]],
    "nested_deep": ["or", ["alw", 1, 9, ["ev", 0, 4, ["not", ["ap", 0]]]],
                    ["and", ["ev", 5, 30, ["ap", 1]], ["alw", -2, 3, ["ap", 2]]]],
    "shared_leaf": ["and", ["alw", 0, 20, ["ap", 0]], ["ev", 0, 20, ["ap", 0]]],
}


def build(spec, lib, cache=None):
    """`cache` maps signal index -> AP object so that a signal used twice is the same leaf (as in build_stl_cache)."""
    cache = {} if cache is None else cache
    kind = spec[0]
    if kind == "ap":
        i = spec[1]
        if i not in cache:
            cache[i] = lib.AP(lambda x, i=i: x[i], comment="s%d" % i)
        return cache[i]
    b = lambda s: build(s, lib, cache)
    if kind == "and":
        return lib.And(b(spec[1]), b(spec[2]))
    if kind == "or":
        return lib.Or(b(spec[1]), b(spec[2]))
    if kind == "not":
        return lib.Not(b(spec[1]))
    if kind == "imply":
        return lib.Imply(b(spec[1]), b(spec[2]))
    if kind == "listand":
        return lib.ListAnd([b(s) for s in spec[1]])
    if kind == "alw":
        return lib.Always(spec[1], spec[2], b(spec[3]))
    if kind == "ev":
        return lib.Eventually(spec[1], spec[2], b(spec[3]))
    if kind == "once":
        return lib.Once(spec[1], spec[2], b(spec[3]))
    if kind == "uu":
        return lib.UntimedUntil(b(spec[1]), b(spec[2]))
    if kind == "until":
        return lib.Until(spec[1], spec[2], b(spec[3]), b(spec[4]))
    raise ValueError(kind)
