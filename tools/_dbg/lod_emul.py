"""CPU emulation of k_octree_head / k_octree_tail's queue logic on the golden octree (stop predicate taken from the reference cut)."""
import numpy as np
g = np.load('tests/golden/lod.npz')
fc, nc = g['oc_first_child'].astype(np.int64), g['oc_n_children'].astype(np.int64)
n = len(nc)
parent = np.full(n, -1, np.int64)
for i in range(n): parent[fc[i]:fc[i] + nc[i]] = i
lv = np.zeros(n, int)
for i in range(1, n): lv[i] = lv[parent[i]] + 1
levels = lv.max() + 1
start = [int(np.searchsorted(lv, l)) for l in range(levels + 1)]
top = 1
while top < levels and top < 8 and start[top + 1] <= 4096: top += 1
walk = min(2, levels - top)
NT = 256
for p in range(3):
    for t in range(len(g['oc_thr'])):
        cut = set(g[f'oc_cut_p{p}_t{t}'].tolist())
        stops = lambda i: i in cut
        emit = set()
        queue = []
        top_end, last_begin = start[top], start[top - 1]
        for wg in range((top_end + 7) // 8):
            q = []
            for node in range(wg * 8, min(wg * 8 + 8, top_end)):
                anc = []; a = parent[node]
                while a != -1: anc.append(a); a = parent[a]
                if any(stops(a) for a in anc): continue
                if stops(node): emit.add(node)
                elif node >= last_begin: q += list(range(fc[node], fc[node] + nc[node]))
            for r in range(walk):
                assert len(q) <= (64 if r == 0 else 512)
                q2 = []
                for i in q:
                    if stops(i): emit.add(i)
                    else: q2 += list(range(fc[i], fc[i] + nc[i]))
                q = q2
            queue += q
        l = top + walk
        below = levels - 1 - l
        cap = 8 * NT * max(below, 1)
        mx = 0
        for b in range((len(queue) + NT - 1) // NT):
            cur = queue[b * NT:(b + 1) * NT]
            stack = []
            while True:
                push = []
                for i in cur:
                    assert lv[i] >= l
                    if stops(i): emit.add(i)
                    else: push += list(range(fc[i], fc[i] + nc[i]))
                stack += push
                mx = max(mx, len(stack)); assert len(stack) <= cap
                if not stack: break
                np_ = min(len(stack), NT)
                cur = stack[len(stack) - np_:]
                del stack[len(stack) - np_:]
        assert emit == cut, (p, t, len(emit), len(cut))
        print(p, t, "ok", len(cut), "queue", len(queue), "max stack", mx, "cap", cap)
