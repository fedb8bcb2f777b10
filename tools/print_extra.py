"""Print the entries of a bench line's `extra` whose key contains argv[1] (bench line on stdin)."""
import json
import sys

line = [l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]
extra = json.loads(line).get("extra", {})
print({k: v for k, v in extra.items() if sys.argv[1] in k})
