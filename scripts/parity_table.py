"""The tolerance table of the GPU parity tests is a FROZEN, reviewed artefact: tests/golden/tolerances.json.

Every GPU comparison is held to 1e-4 of the reference tensor's max-abs (north_star) unless the table lists it.
This tool never widens the table by itself:

    gpurun -- 'MSDF_PARITY_MEASURE=1 python -m pytest tests -m gpu -q'              # writes gpurun_out/parity_errors.json
    python scripts/parity_table.py check   gpurun_out/parity_errors.json [rNN]      # verify + write profiles/rNN_parity_errors.md
    python scripts/parity_table.py tighten gpurun_out/parity_errors.json [rNN]      # LOWER entries to 2 x measured, drop
                                                                                    # entries now under half the bar
    python scripts/parity_table.py add gpurun_out/parity_errors.json rNN 'test|case|key' ['hand-written cause']
    python scripts/parity_table.py prune gpurun_out/parity_errors.json rNN          # drop entries of comparisons that no
                                                                                    # longer exist (their test ran, they did not)
`check` also writes profiles/rNN_parity_admitted.txt: per fp32 row above 1e-4, which yardstick kinds admit it taken alone.
A NEW yardstick kind must be listed in the table's `yardstick_kinds` first (tests/test_host_logic.py refuses kinds that
are not), in a commit of its own: never together with rows it admits.

`check` exits non-zero when a measured error exceeds its tolerance (1e-4 for a comparison not in the table) or when
an entry breaks the rule below.  `tighten` only lowers.  `add` is the one way an entry comes into being: it is
accepted without a cause only inside the rule, and with a hand-written cause otherwise (flagged `hand` in the table,
counted by tests/test_host_logic.py).

Rule for an entry of the fp32 core (the product's default matrix core):
    tol <= max(1e-4, 2 x yardstick)      (+10 % for the rounding of tol to two digits)
where the yardstick is INDEPENDENT of this implementation: how far the REFERENCE's own fp32 result on that tensor
moves (a) against exact arithmetic (fp64; scripts/reference_conditioning.py), (b) when the SDF values its sampler
sees change by 1e-6 relative, (c) by 1e-6 of max|sdf| absolute -- the measured class of difference between two
correct fp32 SDF networks (scripts/reference_sensitivity.py; profiles/r03_reference_sensitivity.json holds all three).
The bf16x6 core (three bf16 planes per operand, six products: fp32-grade) has no rows: its comparisons ('<test>.bf16x6')
are held to the fp32 core's rows; a row of its own is an exception with a hand-written cause.
Entries of the opt-in bf16x3 core (never the default, never in bench.py's `value`) are bounded by
max(8e-4, 2 x 13 x yardstick): that core drops the lo*lo term of every product (2^-16 relative per product), its SDF
values sit 1.2e-5 ... 1.3e-5 of max|sdf| from the reference's (sdf_stages.bf16x3) -- 13 x the 1e-6 the yardstick
perturbs by (first-order scaling of the sampler's sensitivity).
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAR = 1e-4
TABLE = os.path.join(ROOT, 'tests', 'golden', 'tolerances.json')
# round 4 adds the image-mode training case to round 3's file (scripts/reference_sensitivity.py --only ...)
YARD = next(p for p in (os.path.join(ROOT, 'profiles', 'r04_reference_sensitivity.json'),
                        os.path.join(ROOT, 'profiles', 'r03_reference_sensitivity.json')) if os.path.exists(p))
B16_FLOOR = 8e-4
B16_SDF_RATIO = 13.0          # measured |sdf - reference| of the bf16x3 core / the yardstick's 1e-6
OWN_TOL = ('sampler_rounds', 'stages', 'trajectory', 'trajectory.bf16x6')        # tolerances stated in the tests themselves
KINDS = {'fp64': 'against exact (fp64) arithmetic',
         'perturbed_sdf': "when its sampler's SDF values change by 1e-6 relative",
         'perturbed_sdf_abs': "when its sampler's SDF values change by 1e-6 of max|sdf|"}


def yardstick(yard, test, case, key):
    """(largest of the reference's own deviations on this tensor, which one) or (None, None)."""
    ent = yard.get(case)
    if ent is None:
        return None, None
    if test.startswith(('forward_golden', 'sampler_golden')):
        k = 'out.' + key
    elif test.startswith('gradients_golden'):
        k = 'grad.' + key.split('(')[0]
    else:
        return None, None
    best = (None, None)
    for kind in KINDS:
        v = ent.get(kind, {}).get(k)
        if v is not None and (best[0] is None or v > best[0]):
            best = (v, kind)
    return best


def admitting_kinds(yard, test, case, key, tol):
    """Which of the yardstick kinds, each taken ALONE, admit this tolerance under the rule (tol <= max(1e-4, 2 y) + 10 %)."""
    ent = yard.get(case)
    if ent is None:
        return []
    if test.startswith(('forward_golden', 'sampler_golden')):
        k = 'out.' + key
    elif test.startswith('gradients_golden'):
        k = 'grad.' + key.split('(')[0]
    else:
        return []
    return [kind for kind in KINDS if ent.get(kind, {}).get(k) is not None and tol <= bound(test, ent[kind][k])]


def admitted_report(table, yard):
    """Per fp32-core row above the bar: the kinds that admit it.  Returns (lines, rows admitted by (c) alone, rows no kind admits)."""
    lines, c_only, none = [], [], []
    for k, ent in sorted(table.items()):
        test, case, key = k.split('|')
        if 'bf16x3' in test or ent['tol'] <= BAR:
            continue
        kinds = admitting_kinds(yard, test, case, key, ent['tol'])
        lines.append('%-100s tol %.1e  admitted by: %s%s' % (k, ent['tol'], ', '.join(kinds) if kinds else 'none',
                                                            '  [hand-written cause]' if ent.get('hand') else ''))
        if kinds == ['perturbed_sdf_abs']:
            c_only.append(k)
        if not kinds:
            none.append(k)
    return lines, c_only, none


def bound(test, y):
    """Largest tolerance an entry may carry without a hand-written cause."""
    b = max(BAR, 2.0 * (y or 0.0))
    if 'bf16x3' in test:
        b = max(B16_FLOOR, 2.0 * B16_SDF_RATIO * (y or 0.0))
    return b * 1.1


def round_up(x):
    e = math.floor(math.log10(x))
    return math.ceil(x / 10 ** e * 10) / 10 * 10 ** e


def auto_cause(test, y, kind):
    if y is None:
        return 'no reference yardstick for this comparison'
    s = "the reference's own fp32 result on this tensor moves by %.1e %s" % (y, KINDS[kind])
    if 'bf16x3' in test:
        s += '; opt-in bf16x3 core (drops the lo*lo term of every product, DESIGN 4.5)'
    return s


def load_rows(src):
    worst = {}
    for r in json.load(open(src)):
        k = (r['test'], r['case'], r['key'])
        if k not in worst or r['err'] > worst[k][0]:
            worst[k] = (r['err'], r['tol'])
    return worst


def violations(table, yard):
    """Entries that break the rule (used by `check` and by the CPU test)."""
    bad = []
    for k, ent in sorted(table.items()):
        test, case, key = k.split('|')
        y, _ = yardstick(yard, test, case, key)
        if ent['tol'] > bound(test, y) and not ent.get('hand'):
            bad.append((k, ent['tol'], y))
    return bad


def write_md(rnd, worst, table, yard, src):
    md = ['# Measured parity errors on MI355X (%s)\n' % rnd,
          'Written by `scripts/parity_table.py check` from `%s` (the GPU test run with `MSDF_PARITY_MEASURE=1`).' %
          os.path.basename(src),
          'err = max-abs difference / max-abs of the reference tensor (sampler z: / 3.85), reference = the golden',
          'vectors recorded from the imported reference (`*_golden`, `stages`, `sampler_rounds`) or the CPU oracle.',
          'Bar: 1e-4 (north_star).  The table `tests/golden/tolerances.json` is frozen: %d entries; an entry may not exceed'
          % len(table),
          '2 x the reference\'s own deviation on that tensor (yardstick column: the largest of fp32-vs-fp64, SDF values',
          'perturbed by 1e-6 relative, by 1e-6 of max|sdf| absolute) unless it carries a hand-written cause.\n']
    by_test = {}
    for (test, case, key), (err, tol) in sorted(worst.items()):
        by_test.setdefault(test, []).append((case, key, err, tol))
    for test, items in by_test.items():
        md.append('## %s\n' % test)
        md.append('| case | tensor | measured | tolerance | reference yardstick | cause if > 1e-4 |')
        md.append('|---|---|---|---|---|---|')
        for case, key, err, tol in items:
            if test in OWN_TOL:
                md.append('| %s | %s | %.2e | %.1e (stated in the test) | | |' % (case, key, err, tol))
                continue
            ent = table.get('%s|%s|%s' % (test, case, key))
            if ent is None and '.bf16x6' in test:          # held to the fp32 core's row
                ent = table.get('%s|%s|%s' % (test.replace('.bf16x6', '.fp32'), case, key)) or \
                table.get('%s|%s|%s' % (test.replace('.bf16x6', ''), case, key))
            y, kind = yardstick(yard, test, case, key)
            ys = '' if y is None else '%.1e (%s)' % (y, kind)
            if ent:
                md.append('| %s | %s | %.2e | %.1e | %s | %s |' % (case, key, err, ent['tol'], ys,
                                                                  ent['cause'] if ent['tol'] > BAR else ''))
            else:
                md.append('| %s | %s | %.2e | 1e-4 | %s | |' % (case, key, err, ys))
        md.append('')
    with open(os.path.join(ROOT, 'profiles', '%s_parity_errors.md' % rnd), 'w') as f:
        f.write('\n'.join(md))


def save(doc):
    with open(TABLE, 'w') as f:
        json.dump(doc, f, indent=1, sort_keys=True)


def main(argv):
    mode = argv[1] if len(argv) > 1 else 'check'
    src = argv[2] if len(argv) > 2 else os.path.join(ROOT, 'gpurun_out', 'parity_errors.json')
    rnd = argv[3] if len(argv) > 3 else 'r03'
    worst = load_rows(src)
    yard = json.load(open(YARD))['cases']
    doc = json.load(open(TABLE))
    table = doc['tolerances']
    if mode == 'tighten':
        n_low = n_drop = 0
        for k in list(table):
            t = tuple(k.split('|'))
            if t not in worst:
                continue                                   # not measured in this run: left as it is
            err = worst[t][0]
            if err <= BAR / 2:
                del table[k]
                n_drop += 1
            elif round_up(2 * err) < table[k]['tol']:
                table[k]['tol'] = round_up(2 * err)
                table[k]['measured'] = err
                n_low += 1
        save(doc)
        print('tightened %d entries, dropped %d (now %d)' % (n_low, n_drop, len(table)))
    elif mode == 'prune':
        # entries of comparisons that no longer exist: their test family was measured in this (full) run, they were not
        families = {t for (t, _, _) in worst}
        gone = [k for k in table if k.split('|')[0] in families and tuple(k.split('|')) not in worst]
        for k in gone:
            del table[k]
            print('pruned %s' % k)
        save(doc)
    elif mode == 'annotate':
        key, text = argv[4], argv[5]
        table[key]['cause'] = table[key]['cause'].rstrip() + ' -- ' + text
        save(doc)
    elif mode == 'add':
        key = argv[4]
        hand = argv[5] if len(argv) > 5 else None
        t = tuple(key.split('|'))
        err = worst[t][0]
        y, kind = yardstick(yard, *t)
        tol = round_up(2 * err)
        if tol > bound(t[0], y) and not hand:
            print('REFUSED: %s measured %.2e -> tol %.1e exceeds the bound %.1e (yardstick %s); a hand-written cause is '
                  'needed' % (key, err, tol, bound(t[0], y), y))
            return 1
        ent = {'tol': tol, 'measured': err, 'yardstick': y, 'yardstick_kind': kind,
               'cause': hand or auto_cause(t[0], y, kind)}
        if hand:
            ent['hand'] = True
        table[key] = ent
        save(doc)
        print('added %s tol %.1e' % (key, tol))
    # check (also after tighten / add)
    rc = 0
    for (test, case, key), (err, _) in sorted(worst.items()):
        if test in OWN_TOL:
            continue
        ent = table.get('%s|%s|%s' % (test, case, key))
        if ent is None and '.bf16x6' in test:              # the bf16x6 core is held to the fp32 core's rows
            ent = table.get('%s|%s|%s' % (test.replace('.bf16x6', '.fp32'), case, key)) or \
                table.get('%s|%s|%s' % (test.replace('.bf16x6', ''), case, key))
        tol = ent['tol'] if ent else BAR
        if err > tol:
            print('EXCEEDS: %s|%s|%s measured %.2e > %.1e' % (test, case, key, err, tol))
            rc = 1
    for k, tol, y in violations(table, yard):
        print('BREAKS THE RULE: %s tol %.1e, yardstick %s and no hand-written cause' % (k, tol, y))
        rc = 1
    write_md(rnd, worst, table, yard, src)
    lines, c_only, none = admitted_report(table, yard)
    with open(os.path.join(ROOT, 'profiles', '%s_parity_admitted.txt' % rnd), 'w') as f:
        f.write('fp32-core rows of tests/golden/tolerances.json above 1e-4 and the yardstick kinds that admit each, taken '
                'alone\n(fp64 = the reference against exact arithmetic, perturbed_sdf = its sampler\'s SDF values changed by '
                '1e-6 relative,\nperturbed_sdf_abs = by 1e-6 of max|sdf| -- kind (c), introduced in round 3)\n\n')
        f.write('\n'.join(lines) + '\n\n%d rows; %d admitted by (c) perturbed_sdf_abs ALONE; %d by no stored yardstick '
                '(hand-written cause or a comparison without a golden case)\n' % (len(lines), len(c_only), len(none)))
    print('fp32 rows above 1e-4: %d; admitted by yardstick (c) alone: %d; by no stored yardstick: %d  (profiles/%s_parity_admitted.txt)'
          % (len(lines), len(c_only), len(none), rnd))
    f32 = [k for k, e in table.items() if 'bf16x3' not in k and e['tol'] > BAR]
    print('%d comparisons measured; table: %d entries, %d on the fp32 core above 1e-4, %d with a hand-written cause; '
          'worst measured %.2e' % (len(worst), len(table), len(f32), sum(1 for e in table.values() if e.get('hand')),
                                   max(e for (t, _, _), (e, _) in worst.items() if t not in OWN_TOL)))
    return rc


if __name__ == '__main__':
    sys.exit(main(sys.argv))
