#!/bin/bash
# bash scripts/first_contact.sh <dir with the reference's ganon-build and ganon-classify> [--work DIR] [--device N]
# One command that tells whether this repo's filter reader / writer / classifier agree with a real ganon install:
# see scripts/first_contact.py for what is built, cross-loaded, classified and compared.  Needs one MI355X.
cd "$(dirname "$0")/.." && exec python scripts/first_contact.py "$@"
