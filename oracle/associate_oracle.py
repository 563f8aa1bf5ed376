"""TEST INFRASTRUCTURE - CPU restatement of the reference's stamp association
(slam/preprocess/associate.py:70-100) as the literal full cross product: every
(a, b) with |a - (b + offset)| < max_difference, ascending (difference, a, b),
taken greedily while both stamps are unused, result sorted by a.

Parity unpinned: the reference file is Python 2 (list-valued dict.keys(),
cStringIO) and is not run here; its algorithm is deterministic in the inputs
(the key order it iterates in does not survive the two sorts).
"""


def associate(first_list, second_list, offset, max_difference):
    first_keys, second_keys = list(first_list), list(second_list)
    potential = sorted((abs(a - (b + offset)), a, b) for a in first_keys for b in second_keys
                       if abs(a - (b + offset)) < max_difference)
    matches = []
    for _, a, b in potential:
        if a in first_keys and b in second_keys:
            first_keys.remove(a)
            second_keys.remove(b)
            matches.append((a, b))
    return sorted(matches)
