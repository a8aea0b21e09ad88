"""Static scan of a gfx950 .s: spill slots (scratch offsets, AGPRs) whose most
recent STORE before a reload lies inside a conditionally skipped region while
the reload lies after the region's join.

  python tools/spill_hazard/region_scan.py file.s REGION_START REGION_END [POST_END]

REGION_START .. REGION_END: line numbers of the skippable region (from the
`s_cbranch_execz <join>` that skips it to the line before the join label).
When no lane takes the region (execz), a reload after the join reads whatever
the slot held before -- correct only if the value is conditionally defined in
the source as well and the skip path stores its own copy.
"""
import re
import sys


def main():
  path, r0, r1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
  post_end = int(sys.argv[4]) if len(sys.argv) > 4 else r1 + 4000
  lines = open(path).read().split('\n')
  last_store = {}      # slot -> (line, text)
  sreg = {}            # sN -> immediate (s_movk_i32 / s_mov_b32 sN, imm)
  flagged = {}
  for i, l in enumerate(lines[:post_end], 1):
    code = l.split(';')[0].strip()
    if not code or code.startswith('.'):
      continue
    m = re.match(r's_movk?_i32 s(\d+), (0x[0-9a-f]+|\d+)$', code) or \
        re.match(r's_mov_b32 s(\d+), (0x[0-9a-f]+|\d+)$', code)
    if m:
      sreg[int(m.group(1))] = int(m.group(2), 0)
      continue
    m = re.match(r'scratch_(store|load)_dword(x\d)? (.*)$', code)
    if m:
      kind, ops = m.group(1), [o.strip() for o in m.group(3).split(',')]
      off = 0
      mo = re.search(r'offset:(\d+)', code)
      if mo:
        off = int(mo.group(1))
      base = ops[2].split()[0] if kind == 'store' else ops[2].split()[0]
      if base.startswith('s') and base[1:].isdigit():
        off += sreg.get(int(base[1:]), -10**6)
      slots = ['scratch+%d' % off]
    else:
      m = re.match(r'v_accvgpr_(write|read|mov)_b32 (\S+), (\S+)', code)
      if not m:
        continue
      op, dst, src = m.group(1), m.group(2).rstrip(','), m.group(3)
      if op == 'write':
        kind, slots = 'store', [dst]
      elif op == 'read':
        kind, slots = 'load', [src]
      else:
        # a -> a move: a load of src and a store of dst
        s = src
        if s in last_store and r0 <= last_store[s][0] <= r1 and i > r1:
          flagged.setdefault(s, []).append((i, code))
        last_store[dst] = (i, code)
        continue
    for s in slots:
      if kind == 'store':
        last_store[s] = (i, code)
      elif i > r1 and s in last_store and r0 <= last_store[s][0] <= r1:
        flagged.setdefault(s, []).append((i, code))
  for s, uses in sorted(flagged.items(), key=lambda kv: kv[1][0][0]):
    st = last_store[s]
    print('%-14s stored in region at %d (%s); reloaded after the join at %s'
          % (s, st[0], st[1][:50], ', '.join(str(u[0]) for u in uses[:4])))
  print('%d slot(s) reloaded after the join whose latest store is inside the region'
        % len(flagged))


if __name__ == '__main__':
  main()
