#!/bin/bash
# usage: build_variant.sh NAME [extra hipcc flags]   -> variants/NAME.so (developer A/B builds, TG_DEV_MIN)
# Every translation unit of the library is compiled with the flags (any4_amd/build.py, objects under /tmp/tg_variant_obj/NAME/).
set -e
name=$1; shift
cd "$(dirname "$0")/.."; mkdir -p variants
python - "$name" -DTG_DEV_MIN=0 "$@" <<'PY'
import sys, importlib.util, os
spec = importlib.util.spec_from_file_location("b", "any4_amd/build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name, flags = sys.argv[1], sys.argv[2:]
print(b.build(force=True, extra_flags=flags, out=os.path.abspath(f"variants/{name}.so"), obj_dir=f"/tmp/tg_variant_obj/{name}"))
PY
