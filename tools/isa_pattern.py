"""Print the load/store/wait/barrier/MFMA pattern of one kernel from an assembly dump (hipcc -S)."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
a = [i for i, l in enumerate(lines) if l.startswith(key)][0]
body = lines[a:]
end = [i for i, l in enumerate(body) if 's_endpgm' in l][0]
seq = []
for l in body[:end]:
    t = l.strip()
    if 'vmcnt' in t: seq.append('v' + re.search(r'vmcnt\((\d+)\)', t).group(1))
    elif 's_barrier' in t: seq.append('B')
    elif 'global_load_dwordx4' in t: seq.append('L')
    elif 'global_load' in t: seq.append('l')
    elif 'global_store_dwordx4' in t: seq.append('S')
    elif 'global_store' in t: seq.append('s')
    elif re.match(r'^\.LBB\d+_\d+:', t): seq.append('\n' + t.split(':')[0] + ':')
    elif 's_cbranch' in t or 's_branch' in t: seq.append('->' + t.split()[-1])
    elif 'v_mfma' in t: seq.append('M')
out = []
for x in seq:
    if out and out[-1][0] == x: out[-1][1] += 1
    else: out.append([x, 1])
print(' '.join('%s%s' % (x, ('x%d' % n if n > 1 else '')) for x, n in out))
for l in body[:end + 40]:
    if 'vgpr_count' in l or 'NumVgprs' in l or 'ScratchSize' in l or 'Occupancy' in l: print(l.strip())
