"""The C++ mirror is pinned to the reference's PUBLIC headers mechanically: every public method / free function the
reference declares in kfusion/include/kfusion/{warp_field, kinfu, cuda/tsdf_volume, cuda/imgproc, cuda/projective_icp}.hpp must be
declared -- with at least as many overloads -- by the mirror's header of the same name (dynamicfusion_amd/host/include/kfusion/...), or be
listed, with a reason, in INTEGRATION.md's table "Public names of the reference the mirror does not declare".  The reference's headers are
parsed where they lie (nothing is copied); without /root/reference the test skips (the GPU box: nothing to compare with there)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/kfusion/include/kfusion"
MIR = os.path.join(REPO, "dynamicfusion_amd", "host", "include", "kfusion")
HEADERS = ["warp_field.hpp", "kinfu.hpp", "cuda/tsdf_volume.hpp", "cuda/imgproc.hpp", "cuda/projective_icp.hpp"]
KEYWORDS = {"if", "for", "while", "switch", "return", "sizeof", "assert", "defined", "operator", "static_assert", "decltype", "alignas",
            "template", "typedef", "catch"}


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def public_declarations(path):
    """{name: number of declarations} of everything followed by '(' in the public parts of a header (class members after `public:` /
    in a struct, free functions at namespace scope); macros (ALL CAPS) and keywords dropped."""
    names = {}
    access = "public"
    for line in strip_comments(open(path).read()).splitlines():
        l = line.strip()
        m = re.match(r"(class|struct)\s+(?:KF_EXPORTS\s+)?(\w+)", l)
        if m and not l.endswith(";"):
            access = "private" if m.group(1) == "class" else "public"
        if re.match(r"public\s*:", l):
            access = "public"
            continue
        if re.match(r"(private|protected)\s*:", l):
            access = "private"
            continue
        if access != "public":
            continue
        for n in re.findall(r"\b([A-Za-z_]\w*)\s*\(", l):
            if n in KEYWORDS or n.isupper():
                continue
            names[n] = names.get(n, 0) + 1
    return names


def integration_exceptions():
    """names in the INTEGRATION.md table: | `name` | header | reason |"""
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    i = text.find("Public names of the reference the mirror does not declare")
    assert i >= 0, "INTEGRATION.md lost its table of undeclared names"
    out = {}
    for row in text[i:].splitlines()[1:]:
        if row.startswith("#"):
            break
        m = re.match(r"\|\s*`([^`]+)`\s*\|\s*([^|]+)\|\s*([^|]+)\|", row)
        if m:
            out[m.group(1).strip()] = (m.group(2).strip(), m.group(3).strip())
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference not present")
def test_every_public_name_of_the_reference_headers_is_mirrored_or_accounted_for():
    exc = integration_exceptions()
    problems = []
    seen = 0
    for h in HEADERS:
        ref = public_declarations(os.path.join(REF, h))
        mir = public_declarations(os.path.join(MIR, h))
        assert len(ref) >= 5, (h, ref)                      # (the parser found the header's members)
        for name, n in sorted(ref.items()):
            seen += 1
            if mir.get(name, 0) >= n:
                continue
            if name in exc and len(exc[name][1]) > 10:          # listed, with a reason
                continue
            problems.append("%s: %s declared %d time(s) by the reference, %d by the mirror, and not explained in INTEGRATION.md" % (h, name, n, mir.get(name, 0)))
    assert not problems, "\n".join(problems)
    assert seen > 60
    # the table holds nothing stale: every exception is still a name of the reference that the mirror lacks
    for name, (hdr, _) in exc.items():
        ref = public_declarations(os.path.join(REF, hdr))
        mir = public_declarations(os.path.join(MIR, hdr))
        assert name in ref and mir.get(name, 0) < ref[name], "INTEGRATION.md lists `%s` (%s) but the mirror declares it" % (name, hdr)


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference not present")
def test_warp_field_surface_is_complete():
    """The five members VERDICT r5 named, by name: the mirror's WarpField declares them (init(cv::Mat) under KFUSION_USE_OPENCV)."""
    src = strip_comments(open(os.path.join(MIR, "warp_field.hpp")).read())
    for decl in (r"void\s+init\s*\(\s*const\s+cv::Mat\s*&", r"DualQuaternion<float>\s+DQB\s*\(\s*const\s+Vec3f\s*&", r"void\s+getWeightsAndUpdateKNN\s*\(",
                 r"float\s+weighting\s*\(\s*float\s+\w+\s*,\s*float\s+\w+\s*\)\s*const", r"void\s+clear\s*\(\s*\)"):
        assert re.search(decl, src), decl
