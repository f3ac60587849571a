"""Static instruction mix per basic block of one kernel in hipcc's -S output (development aid).
usage: isa_count.py file.s <substring of the mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read()
names = [l.split(':')[0] for l in s.split('\n') if sys.argv[2] in l and re.match(r'^_Z\w+:', l)]
name = names[0]
body = s.split('\n' + name + ':')[1].split('.Lfunc_end')[0]
parts = re.split(r'\n(\.LBB\d+_\d+):', body)
labels = ['entry'] + parts[1::2]
texts = [parts[0]] + parts[2::2]
print(name)
for lab, t in zip(labels, texts):
    ins = [l.strip() for l in t.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//'))]
    f64 = sum(1 for i in ins if re.match(r'v_\w+_f64', i))
    valu = sum(1 for i in ins if i.startswith('v_'))
    ds = sum(1 for i in ins if i.startswith('ds_'))
    vm = sum(1 for i in ins if i.startswith(('global_', 'buffer_', 'scratch_')))
    br = [i.split()[0] + ' ' + i.split()[-1] for i in ins if i.startswith(('s_cbranch', 's_branch'))]
    print(f"{lab:10s} n {len(ins):5d} valu {valu:5d} f64 {f64:5d} ds {ds:4d} vmem {vm:4d} {br}")
