"""Extracts the data rows of the reference's test/test.html (a `dotplot` output committed in the reference repo:
the base-level segments of record 1 of test/testdotplot.paf) into test_html_values.json.
Run in the survey container:  python tests/golden/make_test_html_values.py /root/reference/test/test.html"""
import json, os, re, sys

html = open(sys.argv[1]).read()
m = re.search(r'"data":\{"values":(\[.*?\])\}', html)
values = json.loads(m.group(1))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_html_values.json")
with open(out, "w") as f:
    json.dump(values, f, indent=1, sort_keys=True)
    f.write("\n")
print(len(values), "rows ->", out)
