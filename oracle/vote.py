"""Majority vote of child-row cluster labels per parent group (oracle; test infrastructure only).

Restates `pipeline_modules.py:300-321` (per group: `np.bincount`, `np.argmax` = lowest label among the most frequent,
agree rate = its count / group length, `np.repeat` of the winners) and the group-size distributions of `:332-344`
(`freq_to_prob`, `pipeline_utils.py:1125-1129`) in numpy.  Pinned by tests/golden/g13_group_vote.npz, recorded from
the reference's own statements (tests/golden/make_golden.py::g13_group_vote).
"""
from collections import defaultdict

import numpy as np


def group_vote(cluster_labels, child_group_lengths):
    cluster_labels = np.asarray(cluster_labels)
    group_cluster_labels, agree_rates = [], []
    curr = 0
    for n in child_group_lengths:
        counts = np.bincount(cluster_labels[curr:curr + n])
        group_cluster_labels.append(int(np.argmax(counts)))
        agree_rates.append(np.max(counts) / n)
        curr += n
    group_assignment = np.repeat(group_cluster_labels, child_group_lengths, axis=0).reshape((-1, 1))
    freq = {}
    for lab, n in zip(group_cluster_labels, list(child_group_lengths)):
        freq.setdefault(lab, defaultdict(int))[int(n)] += 1
    prob = {lab: {k: v / sum(list(fd.values())) for k, v in fd.items()} for lab, fd in freq.items()}
    return np.array(group_cluster_labels, dtype=np.int64), np.array(agree_rates), group_assignment, prob
