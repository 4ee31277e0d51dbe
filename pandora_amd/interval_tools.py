"""Regularisation of confidence intervals in ambiguous zones (reference: interval_tools.py:36-96 with
cpp/src/interval_tools.cpp:32-234).  Host-side work on 2-D maps, after the device has reduced the cost volume to the two
bound maps: ambiguous pixels are grouped into horizontal segments, segments that touch on consecutive rows are linked up to
``vertical_depth`` steps away, and every segment takes a quantile of the bounds found in its linked segments."""
import numpy as np


def _segment_neighbours(border_left, border_right):
    """Adjacency lists: segments on consecutive rows whose column spans overlap (interval_tools.cpp:62-77).  The segments come
    sorted by (row, col), as np.argwhere delivers them."""
    n = len(border_left)
    rows = border_left[:, 0]
    row_start = {}
    for i in range(n - 1, -1, -1):
        row_start[int(rows[i])] = i
    adj = [[] for _ in range(n)]
    for i in range(n):
        k = row_start.get(int(rows[i]) + 1)
        if k is None:
            continue
        while k < n and rows[k] == rows[i] + 1:
            if border_left[k, 1] <= border_right[i, 1] and border_right[k, 1] >= border_left[i, 1]:
                adj[i].append(k)
                adj[k].append(i)
            k += 1
    return adj


def create_connected_graph(border_left, border_right, depth):
    """bool [n_segments][n_segments]: entry (i, j) is True when segment j is reached from segment i in at most ``depth`` steps
    between touching segments of consecutive rows; the diagonal is always True (interval_tools.cpp:32-124)."""
    border_left, border_right = np.asarray(border_left), np.asarray(border_right)
    n = len(border_left)
    graph = np.eye(n, dtype=bool)
    if depth == 0 or n == 0:
        return graph
    adj = _segment_neighbours(border_left, border_right)
    for i in range(n):
        seen, frontier = set(adj[i]), set(adj[i])
        for _ in range(1, depth):
            frontier = {y for l in frontier for y in adj[l]} - seen
            if not frontier:
                break
            seen |= frontier
        if seen:
            graph[i, list(seen)] = True
    return graph


def _quantile32(sorted_values, q):
    """Linear-interpolation quantile in float32, the way interval_tools.cpp:203-216 computes it."""
    nb = np.float32(len(sorted_values) - 1)
    q = np.float32(q)
    pos = q * nb
    idx = int(pos)
    if idx >= len(sorted_values) - 1:
        return sorted_values[idx]
    t = pos - np.float32(idx)
    return sorted_values[idx] * (np.float32(1) - t) + sorted_values[idx + 1] * t


def graph_regularization(interval_inf, interval_sup, border_left, border_right, connection_graph, quantile):
    """Every segment i gets, over the pixels of the segments linked to it, the (1 - quantile) quantile of the lower bounds and
    the quantile of the upper bounds (NaN bounds left out; no bound at all -> NaN) -> (inf, sup, mask of regularised pixels)
    (interval_tools.cpp:126-234)."""
    interval_inf, interval_sup = np.asarray(interval_inf, np.float32), np.asarray(interval_sup, np.float32)
    border_left, border_right = np.asarray(border_left), np.asarray(border_right)
    inf_reg, sup_reg = interval_inf.copy(), interval_sup.copy()
    mask = np.zeros(interval_inf.shape, bool)
    p = np.float32(1) - np.float32(quantile)
    for i in range(len(border_left)):
        linked = np.flatnonzero(connection_graph[i])
        lows = [interval_inf[border_left[j, 0], border_left[j, 1]:border_right[j, 1] + 1] for j in linked]
        highs = [interval_sup[border_left[j, 0], border_left[j, 1]:border_right[j, 1] + 1] for j in linked]
        lows = np.sort(np.concatenate(lows)) if lows else np.empty(0, np.float32)
        highs = np.sort(np.concatenate(highs)) if highs else np.empty(0, np.float32)
        lows, highs = lows[~np.isnan(lows)], highs[~np.isnan(highs)]
        if len(lows) > 0 and len(highs) > 0:
            lo, hi = _quantile32(lows, p), _quantile32(highs, quantile)
        else:
            lo = hi = np.float32(np.nan)
        row, c0, c1 = border_left[i, 0], border_left[i, 1], border_right[i, 1]
        inf_reg[row, c0:c1 + 1] = lo
        sup_reg[row, c0:c1 + 1] = hi
        mask[row, c0:c1 + 1] = True
    return inf_reg, sup_reg, mask


def interval_regularization(interval_inf, interval_sup, ambiguity, ambiguity_threshold, ambiguity_kernel_size, vertical_depth=0,
                            quantile_regularization=1.0):
    """interval_tools.py:36-96: a pixel is ambiguous when the minimum of the confidence-from-ambiguity over a horizontal window of
    ``ambiguity_kernel_size`` falls below ``ambiguity_threshold``; runs of ambiguous pixels are the segments."""
    ambiguity = np.asarray(ambiguity)
    n_row, n_col = ambiguity.shape
    pad = ambiguity_kernel_size // 2
    padded = np.hstack((np.ones((n_row, pad)), ambiguity, np.ones((n_row, pad))))
    conf = np.nanmin(np.lib.stride_tricks.sliding_window_view(padded, ambiguity_kernel_size, axis=1), axis=-1)
    conf[:, -1] = 1  # every segment closes inside the row
    steps = np.diff(np.hstack([np.ones((n_row, 1)), conf >= ambiguity_threshold]), axis=-1)
    border_left = np.argwhere(steps == -1)
    border_right = np.argwhere(steps == 1)
    border_right[:, 1] -= 1  # the last ambiguous pixel, not the first confident one
    graph = create_connected_graph(border_left, border_right, vertical_depth)
    return graph_regularization(interval_inf, interval_sup, border_left, border_right, graph, quantile_regularization)
