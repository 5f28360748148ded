"""Edges of a hipGraphDebugDotPrint dump, with node ids renumbered from 0 in creation order: for diffing the captured
training step under different weight-gradient schedules (tools/dbg_defer.py).

    python tools/dot_edges.py a.dot [b.dot]      # one file: nodes with their predecessors; two: the edge difference
"""
import re
import sys


def load(path):
    txt = open(path).read()
    nodes = {}
    for m in re.finditer(r'"graph_\d+_node_(\d+)"\[[^\]]*label="(\d+)\s*\n([^\n"]*)', txt):
        nodes[int(m.group(1))] = m.group(3).strip()
    edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_\d+_node_(\d+)"\s*->\s*"graph_\d+_node_(\d+)"', txt)]
    base = min(nodes)
    names = {k - base: v for k, v in nodes.items()}
    return names, sorted((a - base, b - base) for a, b in edges)


def short(n):
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)(I|Pf|v|PK|RK|\d)", n)
    return m.group(1) if m else n[:40]


if __name__ == "__main__":
    na, ea = load(sys.argv[1])
    if len(sys.argv) == 2:
        preds = {}
        for a, b in ea:
            preds.setdefault(b, []).append(a)
        for k in sorted(na):
            print(k, short(na[k]), "<-", sorted(preds.get(k, [])))
    else:
        nb, eb = load(sys.argv[2])
        print("nodes", len(na), len(nb), "edges", len(ea), len(eb))
        sa, sb = set(ea), set(eb)
        for tag, d, names in (("only in " + sys.argv[1], sa - sb, na), ("only in " + sys.argv[2], sb - sa, nb)):
            print(tag)
            for a, b in sorted(d):
                print("   %d %s -> %d %s" % (a, short(names.get(a, "?")), b, short(names.get(b, "?"))))
