"""Static check of a gfx950 .s file: is every SGPR that frame-index elimination
scavenged for a large scratch offset (`s_movk_i32 sN, imm` + `scratch_* ..., sN`)
dead at that point?  Scans forward in the basic block from each such site: the
first later mention of sN must be a definition, not a use.

  python tools/spill_hazard/scavenge_check.py file.s [kernel]
"""
import re
import sys

NODEST = ('s_cmp', 's_cbranch', 's_branch', 's_waitcnt', 's_nop', 's_barrier',
          's_endpgm', 's_setprio', 's_bitcmp', 's_sendmsg', 's_sleep', 's_setreg',
          's_cselect' + '_never', 'scratch_store', 'global_store', 'ds_write',
          'flat_store', 'buffer_store', 's_setpc', 's_dcache', 's_icache',
          'global_atomic', 'ds_add', 'v_cmpx', 's_code_end')


def regs_in(tok):
  out = set()
  for m in re.finditer(r'\bs\[(\d+):(\d+)\]', tok):
    out.update(range(int(m.group(1)), int(m.group(2)) + 1))
  for m in re.finditer(r'(?<![\w\[])s(\d+)\b', tok):
    out.add(int(m.group(1)))
  return out


def classify(line):
  """(defs, uses) SGPR sets of one instruction line."""
  code = line.split(';')[0].strip()
  if not code or code.startswith('.') or code.endswith(':'):
    return set(), set()
  parts = code.split(None, 1)
  mn = parts[0]
  ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
  # re-join ranges split by the comma inside brackets (none in this syntax)
  defs, uses = set(), set()
  ndest = 1
  if mn.startswith(NODEST):
    ndest = 0
  if mn.startswith(('v_add_co', 'v_sub_co', 'v_addc_co', 'v_subb_co', 'v_subrev_co',
                    'v_div_scale', 'v_mad_u64_u32', 'v_mad_i64_i32')):
    ndest = 2
  for i, o in enumerate(ops):
    r = regs_in(o)
    (defs if i < ndest else uses).update(r)
  return defs, uses


def main():
  path = sys.argv[1]
  kernel = sys.argv[2] if len(sys.argv) > 2 else 'dmc_step'
  lines = open(path).read().split('\n')
  start = lines.index(kernel + ':' + ' '*(40 - len(kernel) - 1) + '; @' + kernel) \
      if False else next(i for i, l in enumerate(lines) if l.startswith(kernel + ':'))
  end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
  sites = flagged = liveout = 0
  for i in range(start, end):
    m = re.match(r'\s*s_movk?_i32 s(\d+), (\S+)', lines[i])
    if not m:
      continue
    n = int(m.group(1))
    nxt = lines[i + 1] if i + 1 < end else ''
    if 'scratch_' not in nxt or not re.search(r', s%d\s*(;|$)' % n, nxt):
      continue
    sites += 1
    j = i + 2
    verdict = 'live-out?'
    while j < end:
      l = lines[j]
      code = l.split(';')[0].strip()
      if code.endswith(':') and code.startswith('.LBB'):
        break
      d, u = classify(l)
      if n in u:
        verdict = 'USE'
        break
      if n in d:
        verdict = 'def'
        break
      if code.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')):
        break
      j += 1
    if verdict == 'USE':
      flagged += 1
      print('line %d: s%d scavenged (%s) but READ at line %d: %s'
            % (i + 1, n, m.group(2), j + 1, lines[j].strip()))
    elif verdict == 'live-out?':
      liveout += 1
  print('%s: %d scavenged-offset sites, %d read-before-def, %d reach the block end '
        'without a mention' % (kernel, sites, flagged, liveout))


if __name__ == '__main__':
  main()
