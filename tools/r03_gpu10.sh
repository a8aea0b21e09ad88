#!/bin/bash
echo "--- f32 then f64 (freed in between)"; timeout -k 10 300 python tools/debug/two_legs.py f32 f64 2>&1 | grep "ms per"
echo "--- f64 then f32"; timeout -k 10 300 python tools/debug/two_legs.py f64 f32 2>&1 | grep "ms per"
echo "--- f32 then f32"; timeout -k 10 300 python tools/debug/two_legs.py f32 f32 2>&1 | grep "ms per"
echo "--- f64 f64"; timeout -k 10 300 python tools/debug/two_legs.py f64 f64 2>&1 | grep "ms per"
