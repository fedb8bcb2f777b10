#!/bin/bash
# round 6, call 18: call 16 + call 17 in one (parity of the faces-from-arrays sweep, walls timing, kernel trace)
bash tools/r6_calls/call16.sh 2>/dev/null | grep -v "^finished"
bash tools/r6_calls/call17.sh
