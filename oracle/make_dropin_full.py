#!/usr/bin/env python3
"""Build aid for oracle/_ref/liborbslam_dropin_full.so: the OPTIONAL steps of INTEGRATION.md §2 (3b, 3d') applied to the reference's
src/Frame.cc.  Reads the reference source where it lies, replaces the BODIES of four Frame members by the one-line forwards to the
drop-in extractor that INTEGRATION.md shows, and writes the result to the path given (a temporary file the Makefile deletes after
compiling it — no reference source is kept in this repository).  usage: make_dropin_full.py <Frame.cc> <out.cc>"""
import re
import sys

FORWARDS = {
    # the only reader of mvImagePyramid becomes a device call on both extractors' resident results
    r"void\s+Frame::ComputeStereoMatches\s*\(\s*\)":
        "{ mpORBextractorLeft->ComputeStereoMatches(*mpORBextractorRight, mbf, mb, N, mvuRight, mvDepth); }",
    # §2-3d': mvKeysUn / the image bounds / the depth lookup come from the extractor that just processed this frame
    r"void\s+Frame::UndistortKeyPoints\s*\(\s*\)":
        "{ mpORBextractorLeft->UndistortKeyPoints(mvKeysUn); }",
    r"void\s+Frame::ComputeImageBounds\s*\(\s*const\s+cv::Mat\s*&\s*imLeft\s*\)":
        "{ mpORBextractorLeft->ComputeImageBounds(imLeft.cols, imLeft.rows, mnMinX, mnMaxX, mnMinY, mnMaxY); }",
    r"void\s+Frame::ComputeStereoFromRGBD\s*\(\s*const\s+cv::Mat\s*&\s*imDepth\s*\)":
        "{ mpORBextractorLeft->ComputeStereoFromRGBD(imDepth, 1.0f, mbf, N, mvuRight, mvDepth); }",
}


def replace_body(src, signature, body):
    m = re.search(signature, src)
    if not m:
        raise SystemExit(f"signature not found: {signature}")
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        c = src[j]
        depth += c == "{"
        depth -= c == "}"
        j += 1
        if depth == 0:
            break
    return src[:i] + body + src[j:]


def main():
    src = open(sys.argv[1]).read()
    for sig, body in FORWARDS.items():
        src = replace_body(src, sig, body)
    open(sys.argv[2], "w").write(src)


if __name__ == "__main__":
    main()
