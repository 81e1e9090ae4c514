// Stand-in for <opencv2/core/core.hpp>, used ONLY to compile the reference's own DBoW2 sources (Thirdparty/DBoW2, vendored in
// /root/reference) into oracle/_ref/libdbow2_ref.so, the real-reference oracle of the BoW row.  OpenCV is absent from this
// image; DBoW2's vocabulary transform / text I/O touch cv::Mat only as a ref-counted byte container (cvlite's Mat has the same
// shallow-copy semantics) and cv::FileStorage only in the YAML save/load members, which are virtual (so they must compile)
// but are never called by ORB_SLAM2 (System.cc:68 uses loadFromTextFile).  The FileStorage stubs abort if reached.
// Test infrastructure, not product code.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <cmath>
#include <cfloat>
#include <climits>
#include <cstring>
#include <algorithm>
#include <iostream>
#include <sstream>
#include <vector>
#include "cvlite/cvlite.h"   // the real header pulls these standard headers in too; DBoW2 relies on that

namespace cv {

class FileNode {
public:
    size_t size() const { die(); return 0; }
    FileNode operator[](int) const { die(); return FileNode(); }
    FileNode operator[](const char*) const { die(); return FileNode(); }
    FileNode operator[](const std::string&) const { die(); return FileNode(); }
    operator int() const { die(); return 0; }
    operator double() const { die(); return 0; }
    operator std::string() const { die(); return std::string(); }
private:
    static void die() { fprintf(stderr, "cv::FileStorage is not part of the oracle build (YAML vocabulary I/O is unused by ORB_SLAM2)\n"); abort(); }
};

class FileStorage {
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage(const char*, int) {}
    FileStorage(const std::string&, int) {}
    bool isOpened() const { return false; }
    FileNode operator[](const std::string&) const { return FileNode(); }
    FileNode operator[](const char*) const { return FileNode(); }
};
template <typename T> inline FileStorage& operator<<(FileStorage& fs, const T&) { return fs; }

}  // namespace cv
