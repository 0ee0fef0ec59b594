"""Build check: no kernel whose name contains PREFIX may use scratch (register spills).
usage: check_noscratch.py REMARKS_FILE PREFIX [ALLOW,ALLOW...]  (diagnostic builds only: kernels whose
name contains an ALLOW substring may spill -- they must then never be launched)   (REMARKS_FILE = hipcc -Rpass-analysis=kernel-resource-usage output)"""
import re, sys
name, bad, seen = None, [], 0
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
    m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
    if m and name and sys.argv[2] in name:
        seen += 1
        if int(m.group(1)):
            bad.append((name, int(m.group(1))))
if not seen:
    sys.exit("check_noscratch: no %s kernels found in %s" % (sys.argv[2], sys.argv[1]))
allow = sys.argv[3].split(',') if len(sys.argv) > 3 and sys.argv[3] else []
for n, b in bad:
    if any(a in n for a in allow):
        print('check_noscratch: WARNING %s spills %d bytes/lane (allowed: never launch it)' % (n, b))
bad = [(n, b) for n, b in bad if not any(a in n for a in allow)]
if bad:
    for n, b in bad:
        print("check_noscratch: %s spills %d bytes/lane" % (n, b))
    sys.exit(1)
print("check_noscratch: %d %s kernels, none uses scratch" % (seen, sys.argv[2]))
