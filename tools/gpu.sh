#!/bin/bash
# gpurun wrapper: the snapshot pushed to the GPU box has no .git, so the commit the binaries were built from travels in
# .git_rev (git-ignored); profile summaries generated on the box copy it into their "git" field.
cd "$(dirname "$0")/.." && { git rev-parse --short HEAD | tr -d '\n'; git diff --quiet HEAD -- . ':!profiles' || printf '+dirty'; } > .git_rev
exec /usr/local/graft/bin/gpurun "$@"
