#!/usr/bin/env python3
"""INTEGRATION.md shows bindings/rust/{build.rs, src/ffi.rs, src/mod.rs} verbatim (tests/test_abi_and_host.py keeps them
identical): after editing a binding file, run this to put its text back into the document's sections 2 - 5."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc_path = os.path.join(ROOT, "INTEGRATION.md")
doc = open(doc_path).read()
for heading, rel in (("## 2.", "build.rs"), ("## 3.", os.path.join("src", "ffi.rs")), ("## 4.", os.path.join("src", "mod.rs")),
                     ("## 5.", os.path.join("src", "tests.rs"))):
    text = open(os.path.join(ROOT, "bindings", "rust", rel)).read().rstrip("\n")
    at = doc.index(heading)
    m = re.compile(r"```rust\n.*?\n```", re.S).search(doc, at)
    doc = doc[:m.start()] + "```rust\n" + text + "\n```" + doc[m.end():]
open(doc_path, "w").write(doc)
print("INTEGRATION.md synchronised with bindings/rust")
