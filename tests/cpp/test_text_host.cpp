// Host-only tests of include/minnow_text.hpp (no GPU needed), written after the reference's go/text/text_test.go:
// TestLineHeader (21-39), TestCommentHeader (41-61), the block chain of TestNextBlock (151-178) and TestRockstarHeader (180-208).
// argv[1] = scratch dir.  Run by tests/test_cpp_host.py.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "minnow_text.hpp"

static int failures = 0;
#define EXPECT(cond, msg) do { if (!(cond)) { std::cerr << "FAIL " << __func__ << ": " << msg << "\n"; ++failures; } } while (0)
static std::string scratch;

static std::string writeFile(const std::string& name, const std::string& text) {
    const std::string p = scratch + "/" + name;
    std::ofstream f(p, std::ios::binary);
    f << text;
    return p;
}
static text::ReaderConfig cfg(int itemSize, int blockSize) { text::ReaderConfig c; c.MaxItemSize = itemSize; c.MaxBlockSize = blockSize; return c; }

static void TestLineHeader() {
    struct { const char* text; int itemSize, lines; const char* res; } tests[] = {{"a\nb\nc\nd", 100, 2, "a\nb"}, {"a\nb\nc\nd", 5, 2, "a\nb"}};
    int i = 0;
    for (auto& t : tests) {
        text::Reader rd(writeFile("line_header.txt", t.text), cfg(t.itemSize, 1000));
        const std::string out = rd.LineHeader(t.lines);
        EXPECT(out == t.res, "Test " << i << ": expected '" << t.res << "', got '" << out << "'.");
        ++i;
    }
}

static void TestCommentHeader() {
    struct { const char* text; int itemSize; const char* res; } tests[] = {{"#a\n#b\nc\nd", 100, "#a\n#b"}, {"#a\n#b\nc\nd", 8, "#a\n#b"}};
    int i = 0;
    for (auto& t : tests) {
        text::Reader rd(writeFile("comment_header.txt", t.text), cfg(t.itemSize, 1000));
        const std::string out = rd.CommentHeader();
        EXPECT(out == t.res, "Test " << i << ": expected '" << t.res << "', got '" << out << "'.");
        ++i;
    }
}

// the chain nextBlock produces from the start of TestNextBlock's file: next = pos + 12 - 6 rounded up to a line start, -1 once
// pos + 12 >= 30
static void TestBlocks() {
    const std::string text = "1234\n1234\n1234\n1234\n1234\n1234\n";
    text::Reader rd(writeFile("next_block.txt", text), cfg(6, 12));
    EXPECT(rd.Blocks() == 3, "blocks " << rd.Blocks());
    for (int b = 0; b < rd.Blocks(); ++b) EXPECT(rd.BlockSize(b) == 10, "block " << b << " has " << rd.BlockSize(b) << " bytes");
    // one block when the file fits; MaxBlockSize / 2 < MaxItemSize panics (text.go:55-59)
    text::Reader one(writeFile("one_block.txt", text), cfg(10, 1000));
    EXPECT(one.Blocks() == 1 && one.BlockSize(0) == 30, "a single block");
    bool threw = false;
    try { text::Reader bad(writeFile("bad.txt", text), cfg(10, 12)); } catch (const text::Error&) { threw = true; }
    EXPECT(threw, "MaxBlockSize / 2 < MaxItemSize must throw");
    threw = false;
    try { text::Reader none(scratch + "/does_not_exist.txt"); } catch (const text::Error&) { threw = true; }
    EXPECT(threw, "a missing file must throw");
    threw = false;
    try { text::Reader rd2(writeFile("nonl.txt", std::string(40, 'x')), cfg(6, 12)); } catch (const text::Error&) { threw = true; }
    EXPECT(threw, "no newline within MaxItemSize must throw");
}

static void TestRockstarHeader() {
    const std::string rockstarLine = "#ID DescID Mvir Vmax Vrms Rvir Rs Np X Y Z VX VY VZ JX JY JZ Spin rs_klypin Mvir_all M200b M200c M500c M2500c Xoff Voff spin_bullock b_to_a c_to_a A[x] A[y] A[z] b_to_a(500c) c_to_a(500c) A[x](500c) A[y](500c) A[z](500c) T/|U| M_pe_Behroozi M_pe_Diemer Halfmass_Radius rvmax PID";
    const std::string consistentTreesLine = "#scale(0) id(1) desc_scale(2) desc_id(3) num_prog(4) pid(5) upid(6) desc_pid(7) phantom(8) sam_Mvir(9) Mvir(10) Rvir(11) rs(12) vrms(13) mmp?(14) scale_of_last_MM(15) vmax(16) x(17) y(18) z(19) vx(20) vy(21) vz(22) Jx(23) Jy(24) Jz(25) Spin(26) Breadth_first_ID(27) b_to_a(500c)(51) A[x](500c)(53) T/|U|(56) Acc_Rate_1*Tdyn(67) Vmax\\@Mpeak(77) Log_(Vmax/Vmax_max(Tdyn;Tmpeak))(79) Future_merger_MMP_ID(81)";
    const std::vector<std::string> rockstarExp = {"ID", "DescID", "Mvir", "Vmax", "Vrms", "Rvir", "Rs", "Np", "X", "Y", "Z", "VX", "VY", "VZ", "JX", "JY", "JZ", "Spin", "rs_klypin", "Mvir_all", "M200b", "M200c", "M500c", "M2500c", "Xoff", "Voff", "spin_bullock", "b_to_a", "c_to_a", "A[x]", "A[y]", "A[z]", "b_to_a(500c)", "c_to_a(500c)", "A[x](500c)", "A[y](500c)", "A[z](500c)", "T/|U|", "M_pe_Behroozi", "M_pe_Diemer", "Halfmass_Radius", "rvmax", "PID"};
    const std::vector<std::string> consistentTreesExp = {"scale", "id", "desc_scale", "desc_id", "num_prog", "pid", "upid", "desc_pid", "phantom", "sam_Mvir", "Mvir", "Rvir", "rs", "vrms", "mmp?", "scale_of_last_MM", "vmax", "x", "y", "z", "vx", "vy", "vz", "Jx", "Jy", "Jz", "Spin", "Breadth_first_ID", "b_to_a(500c)", "A[x](500c)", "T/|U|", "Acc_Rate_1*Tdyn", "Vmax\\@Mpeak", "Log_(Vmax/Vmax_max(Tdyn;Tmpeak))", "Future_merger_MMP_ID"};
    EXPECT(text::Rockstar::rockstarNames(rockstarLine) == rockstarExp, "rockstar names");
    EXPECT(text::Rockstar::consistentTreesNames(consistentTreesLine) == consistentTreesExp, "consistent-trees names");
    // Names() picks the flavour by "(0)" in the first line (rockstar.go:16-23)
    text::Rockstar r1(writeFile("rs.txt", rockstarLine + "\n#second\n1 2 3\n"));
    EXPECT(r1.Names() == rockstarExp, "Names() of a Rockstar header");
    EXPECT(r1.Header() == rockstarLine + "\n#second", "Header() = the comment lines");
    text::Rockstar r2(writeFile("ct.txt", consistentTreesLine + "\n1 2 3\n"));
    EXPECT(r2.Names() == consistentTreesExp, "Names() of a consistent-trees header");
}

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: test_text_host <scratch dir>\n"; return 2; }
    scratch = argv[1];
    try {
        TestLineHeader();
        TestCommentHeader();
        TestBlocks();
        TestRockstarHeader();
    } catch (const std::exception& e) {
        std::cerr << "EXCEPTION: " << e.what() << "\n";
        return 1;
    }
    if (failures) { std::cerr << failures << " failure(s)\n"; return 1; }
    std::cout << "ok: 4 text host tests\n";
    return 0;
}
