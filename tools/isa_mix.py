"""instruction mix of one kernel of a hipcc -S dump: python tools/isa_mix.py file.s <mangled-name-substring> [--scratch]"""
import collections
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
starts = [i for i in range(len(s)) if s.startswith("\n_Z", i) and name in s[i:s.index(":", i)]]
i = starts[0]
j = s.index(".end_amdhsa_kernel", i)
lines = s[i:j].split("\n")
c = collections.Counter()
for l in lines:
    l = l.strip()
    if not l or l.startswith((".", ";")):
        continue
    c[l.split()[0]] += 1
print(len(lines), "lines")
for op, n in c.most_common(45):
    print(f"{op:40s} {n}")
if "--scratch" in sys.argv:
    for n, l in enumerate(lines):
        if "scratch_" in l:
            print(n, l.strip())
