// gfx950emu (test infrastructure, see emu.h): the code object -- ELF segments, kernel descriptors, the msgpack metadata note -- and its
// .text as llvm-objdump prints it, parsed into Inst records.
#include "emu.h"
#include <elf.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include <functional>
#include <map>

static const char *const g_op_names[] = {
#define X(n) #n,
    OPS(X)
#undef X
};
const char *op_name(u16 op) { return op < OP_COUNT ? g_op_names[op] : "?"; }

static std::unordered_map<std::string, u16> &op_map()
{
    static std::unordered_map<std::string, u16> m;
    if (m.empty()) for (u16 i = 0; i < OP_COUNT; i++) m[g_op_names[i]] = i;
    return m;
}
int op_lookup(const std::string &mn, u8 *enc)
{
    auto &m = op_map();
    *enc = E_PLAIN;
    auto it = m.find(mn);
    if (it != m.end()) return it->second;
    static const struct { const char *suf; u8 enc; } sufs[] = {{"_e32", E_PLAIN}, {"_e64", E_E64}, {"_sdwa", E_SDWA}, {"_dpp", E_DPP}};
    for (auto &s : sufs) {
        const size_t l = strlen(s.suf);
        if (mn.size() > l && mn.compare(mn.size() - l, l, s.suf) == 0) {
            it = m.find(mn.substr(0, mn.size() - l));
            if (it != m.end()) { *enc = s.enc; return it->second; }
        }
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------- msgpack (the subset)
struct MP {
    enum T { NIL, BOOL, INT, STR, ARR, MAP, BIN } t = NIL;
    i64 i = 0;
    std::string s;
    std::vector<MP> a;                      // array, or map as key, value, key, value
    const MP *get(const char *key) const { if (t != MAP) return nullptr; for (size_t k = 0; k + 1 < a.size(); k += 2) if (a[k].s == key) return &a[k + 1]; return nullptr; }
};
static bool mp_read(const u8 *&p, const u8 *end, MP &o)
{
    if (p >= end) return false;
    const u8 b = *p++;
    auto be = [&](int n) -> u64 { u64 v = 0; for (int k = 0; k < n; k++) v = (v << 8) | *p++; return v; };
    auto str = [&](size_t n) { o.t = MP::STR; o.s.assign((const char *)p, n); p += n; return true; };
    auto arr = [&](size_t n, bool map) { o.t = map ? MP::MAP : MP::ARR; o.a.resize(map ? 2 * n : n); for (auto &e : o.a) if (!mp_read(p, end, e)) return false; return true; };
    if (b <= 0x7f) { o.t = MP::INT; o.i = b; return true; }
    if (b >= 0xe0) { o.t = MP::INT; o.i = (i8)b; return true; }
    if ((b & 0xe0) == 0xa0) return str(b & 0x1f);
    if ((b & 0xf0) == 0x90) return arr(b & 0x0f, false);
    if ((b & 0xf0) == 0x80) return arr(b & 0x0f, true);
    switch (b) {
    case 0xc0: o.t = MP::NIL; return true;
    case 0xc2: case 0xc3: o.t = MP::BOOL; o.i = b & 1; return true;
    case 0xc4: { size_t n = be(1); o.t = MP::BIN; p += n; return true; }
    case 0xc5: { size_t n = be(2); o.t = MP::BIN; p += n; return true; }
    case 0xc6: { size_t n = be(4); o.t = MP::BIN; p += n; return true; }
    case 0xcc: o.t = MP::INT; o.i = (i64)be(1); return true;
    case 0xcd: o.t = MP::INT; o.i = (i64)be(2); return true;
    case 0xce: o.t = MP::INT; o.i = (i64)be(4); return true;
    case 0xcf: o.t = MP::INT; o.i = (i64)be(8); return true;
    case 0xd0: o.t = MP::INT; o.i = (i8)be(1); return true;
    case 0xd1: o.t = MP::INT; o.i = (i16)be(2); return true;
    case 0xd2: o.t = MP::INT; o.i = (i32)be(4); return true;
    case 0xd3: o.t = MP::INT; o.i = (i64)be(8); return true;
    case 0xd9: return str(be(1));
    case 0xda: return str(be(2));
    case 0xdb: return str(be(4));
    case 0xdc: return arr(be(2), false);
    case 0xdd: return arr(be(4), false);
    case 0xde: return arr(be(2), true);
    case 0xdf: return arr(be(4), true);
    case 0xca: p += 4; o.t = MP::NIL; return true;
    case 0xcb: p += 8; o.t = MP::NIL; return true;
    }
    fprintf(stderr, "gfx950emu: msgpack byte 0x%02x not handled\n", b);
    return false;
}

// ---------------------------------------------------------------------------------------------------------------- ELF
bool load_code_object(CodeObject &co, const u8 *elf, size_t size)
{
    co.elf.assign(elf, elf + size);
    const Elf64_Ehdr *eh = (const Elf64_Ehdr *)co.elf.data();
    if (size < sizeof *eh || memcmp(eh->e_ident, ELFMAG, 4) != 0) { fprintf(stderr, "gfx950emu: not an ELF code object\n"); return false; }
    const Elf64_Phdr *ph = (const Elf64_Phdr *)(co.elf.data() + eh->e_phoff);
    u64 top = 0;
    for (int k = 0; k < eh->e_phnum; k++) if (ph[k].p_type == PT_LOAD && ph[k].p_vaddr + ph[k].p_memsz > top) top = ph[k].p_vaddr + ph[k].p_memsz;
    co.image.assign(top + 256, 0);
    for (int k = 0; k < eh->e_phnum; k++) if (ph[k].p_type == PT_LOAD) memcpy(co.image.data() + ph[k].p_vaddr, co.elf.data() + ph[k].p_offset, ph[k].p_filesz);
    const Elf64_Shdr *sh = (const Elf64_Shdr *)(co.elf.data() + eh->e_shoff);
    // symbols: kernels (FUNC) and their descriptors (<name>.kd)
    std::unordered_map<std::string, u64> kd, fn;
    for (int k = 0; k < eh->e_shnum; k++) {
        if (sh[k].sh_type != SHT_SYMTAB && sh[k].sh_type != SHT_DYNSYM) continue;
        const Elf64_Sym *sy = (const Elf64_Sym *)(co.elf.data() + sh[k].sh_offset);
        const char *strs = (const char *)(co.elf.data() + sh[sh[k].sh_link].sh_offset);
        for (size_t i = 0; i < sh[k].sh_size / sizeof *sy; i++) {
            const std::string n = strs + sy[i].st_name;
            if (n.size() > 3 && n.compare(n.size() - 3, 3, ".kd") == 0) kd[n.substr(0, n.size() - 3)] = sy[i].st_value;
            else if (ELF64_ST_TYPE(sy[i].st_info) == STT_FUNC) fn[n] = sy[i].st_value;
        }
    }
    for (auto &e : kd) {
        KernelInfo ki;
        ki.name = e.first;
        const u8 *d = co.image.data() + e.second;
        memcpy(&ki.lds_static, d + 0, 4); memcpy(&ki.scratch, d + 4, 4); memcpy(&ki.kernarg_size, d + 8, 4);
        i64 entry_off; memcpy(&entry_off, d + 16, 8);
        ki.entry = e.second + entry_off;
        memcpy(&ki.rsrc3, d + 44, 4); memcpy(&ki.rsrc1, d + 48, 4); memcpy(&ki.rsrc2, d + 52, 4);
        u16 props; memcpy(&props, d + 56, 2); ki.props = props;
        co.kernels[e.first] = ki;
    }
    // metadata note: arguments, register counts
    for (int k = 0; k < eh->e_phnum; k++) {
        if (ph[k].p_type != PT_NOTE) continue;
        const u8 *p = co.elf.data() + ph[k].p_offset, *end = p + ph[k].p_filesz;
        while (p + 12 <= end) {
            u32 nsz, dsz, type; memcpy(&nsz, p, 4); memcpy(&dsz, p + 4, 4); memcpy(&type, p + 8, 4);
            const u8 *name = p + 12, *desc = name + ((nsz + 3) & ~3u);
            if (type == 32 && nsz >= 6 && memcmp(name, "AMDGPU", 6) == 0) {
                MP root; const u8 *q = desc;
                if (!mp_read(q, desc + dsz, root)) return false;
                const MP *ks = root.get("amdhsa.kernels");
                if (ks) for (auto &km : ks->a) {
                    const MP *nm = km.get(".name");
                    if (!nm) continue;
                    auto it = co.kernels.find(nm->s);
                    if (it == co.kernels.end()) continue;
                    KernelInfo &ki = it->second;
                    if (auto *v = km.get(".vgpr_count")) ki.vgprs = (u32)v->i;
                    if (auto *v = km.get(".agpr_count")) ki.agprs = (u32)v->i;
                    if (auto *v = km.get(".sgpr_count")) ki.sgprs = (u32)v->i;
                    if (auto *as = km.get(".args")) for (auto &am : as->a) {
                        KernArg a; a.offset = (u32)am.get(".offset")->i; a.size = (u32)am.get(".size")->i;
                        a.kind = am.get(".value_kind") ? am.get(".value_kind")->s : "";
                        ki.args.push_back(a);
                    }
                }
            }
            p = desc + ((dsz + 3) & ~3u);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------- disassembly text
static std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
static bool parse_reg_range(const std::string &s, size_t pos, u16 *reg, u16 *n)      // "12" or "[4:7]" from pos
{
    if (pos >= s.size()) return false;
    if (s[pos] == '[') {
        unsigned a, b;
        if (sscanf(s.c_str() + pos, "[%u:%u]", &a, &b) != 2) return false;
        *reg = (u16)a; *n = (u16)(b - a + 1);
        return true;
    }
    if (!isdigit((unsigned char)s[pos])) return false;
    *reg = (u16)atoi(s.c_str() + pos); *n = 1;
    return true;
}
static bool parse_operand(std::string t, Opnd &o)
{
    o = Opnd();
    t = trim(t);
    if (t.empty()) return false;
    if (t.compare(0, 5, "sext(") == 0 && t.back() == ')') { o.flags |= F_SEXT; t = t.substr(5, t.size() - 6); }
    bool neg = false;
    if (t[0] == '-' && t.size() > 1 && !isdigit((unsigned char)t[1]) && t[1] != '.') { neg = true; t = t.substr(1); }
    if (t.compare(0, 5, "sext(") == 0 && t.back() == ')') { o.flags |= F_SEXT; t = t.substr(5, t.size() - 6); }
    if (t[0] == '|' && t.back() == '|') { o.flags |= F_ABS; t = t.substr(1, t.size() - 2); }
    if (t.compare(0, 4, "neg(") == 0 && t.back() == ')') { neg = !neg; t = t.substr(4, t.size() - 5); }
    if (t.compare(0, 4, "abs(") == 0 && t.back() == ')') { o.flags |= F_ABS; t = t.substr(4, t.size() - 5); }
    if (neg) o.flags |= F_NEG;
    static const std::map<std::string, u8> named = {
        {"vcc", K_VCC}, {"vcc_lo", K_VCC_LO}, {"vcc_hi", K_VCC_HI}, {"exec", K_EXEC}, {"exec_lo", K_EXEC_LO}, {"exec_hi", K_EXEC_HI},
        {"m0", K_M0}, {"scc", K_SCC}, {"src_scc", K_SCC}, {"off", K_OFF}, {"src_shared_base", K_SHARED_BASE}, {"src_private_base", K_PRIVATE_BASE},
        {"src_shared_limit", K_SHARED_LIMIT}, {"src_private_limit", K_PRIVATE_LIMIT}, {"null", K_NULL}};
    auto it = named.find(t);
    if (it != named.end()) { o.kind = it->second; o.n = (o.kind == K_VCC || o.kind == K_EXEC) ? 2 : 1; return true; }
    if ((t[0] == 'v' || t[0] == 's' || t[0] == 'a') && t.size() > 1 && (isdigit((unsigned char)t[1]) || t[1] == '[')) {
        if (!parse_reg_range(t, 1, &o.reg, &o.n)) return false;
        o.kind = t[0] == 'v' ? K_VGPR : t[0] == 's' ? K_SGPR : K_AGPR;
        return true;
    }
    // numbers
    const char *c = t.c_str();
    char *e = nullptr;
    if (t.find("0x") == 0 || t.find("-0x") == 0) { o.kind = K_IMM; o.imm = (i64)strtoull(c + (c[0] == '-' ? 1 : 0), &e, 16); if (c[0] == '-') o.imm = -o.imm; return *e == 0; }
    if (t.find('.') != std::string::npos || t.find('e') != std::string::npos) { o.kind = K_FIMM; o.f = strtod(c, &e); return *e == 0; }
    if (isdigit((unsigned char)c[0]) || c[0] == '-') { o.kind = K_IMM; o.imm = strtoll(c, &e, 10); return *e == 0; }
    return false;
}
static u8 parse_sel(const std::string &v)
{
    if (v == "BYTE_0") return SEL_BYTE0; if (v == "BYTE_1") return SEL_BYTE1; if (v == "BYTE_2") return SEL_BYTE2; if (v == "BYTE_3") return SEL_BYTE3;
    if (v == "WORD_0") return SEL_WORD0; if (v == "WORD_1") return SEL_WORD1;
    return SEL_DWORD;
}

static bool parse_line(const std::string &line, Inst &in, std::string &err)
{
    // "\tmnemonic operands   // ADDR: ENCODING <sym+off>"
    const size_t cpos = line.find("//");
    if (cpos == std::string::npos) return false;
    const std::string body = trim(line.substr(0, cpos));
    unsigned long long addr = 0;
    if (sscanf(line.c_str() + cpos + 2, " %llx:", &addr) != 1) return false;
    in = Inst();
    in.addr = addr;
    size_t sp = body.find_first_of(" \t");
    const std::string mn = sp == std::string::npos ? body : body.substr(0, sp);
    std::string rest = sp == std::string::npos ? "" : trim(body.substr(sp));
    const int op = op_lookup(mn, &in.enc);
    if (op < 0) { err = "unknown mnemonic " + mn; return false; }
    in.op = (u16)op;
    if (in.op == OP_s_waitcnt || in.op == OP_s_nop || in.op == OP_s_endpgm || in.op == OP_s_barrier || in.op == OP_s_code_end ||
        in.op == OP_buffer_inv || in.op == OP_buffer_wbl2 || in.op == OP_buffer_wbinvl1 || in.op == OP_s_dcache_wb || in.op == OP_s_icache_inv ||
        in.op == OP_s_dcache_inv || in.op == OP_s_set_gpr_idx_off || in.op == OP_s_setprio || in.op == OP_v_nop || in.op == OP_ds_nop) return true;
    if (in.op == OP_s_set_gpr_idx_on) {
        // s_set_gpr_idx_on s62, gpr_idx(DST)   |  gpr_idx(SRC0,DST)
        const size_t c = rest.find(',');
        if (!parse_operand(rest.substr(0, c), in.o[0])) { err = "operand"; return false; }
        in.no = 1;
        if (rest.find("SRC0") != std::string::npos) in.gpr_idx_mode |= 1;
        if (rest.find("SRC1") != std::string::npos) in.gpr_idx_mode |= 2;
        if (rest.find("SRC2") != std::string::npos) in.gpr_idx_mode |= 4;
        if (rest.find("DST") != std::string::npos) in.gpr_idx_mode |= 8;
        return true;
    }
    // split at commas outside brackets / parentheses; tokens behind the first of a piece are modifiers
    std::vector<std::string> pieces;
    {
        int depth = 0; std::string cur;
        for (char ch : rest) {
            if (ch == '[' || ch == '(') depth++;
            if (ch == ']' || ch == ')') depth--;
            if (ch == ',' && depth == 0) { pieces.push_back(cur); cur.clear(); } else cur += ch;
        }
        if (!trim(cur).empty()) pieces.push_back(cur);
    }
    std::vector<std::string> mods;
    for (auto &pc : pieces) {
        std::vector<std::string> toks;
        { int depth = 0; std::string cur;
          for (char ch : trim(pc)) {
              if (ch == '[' || ch == '(' ) depth++;
              if (ch == ']' || ch == ')') depth--;
              if (isspace((unsigned char)ch) && depth == 0) { if (!cur.empty()) toks.push_back(cur); cur.clear(); } else cur += ch;
          }
          if (!cur.empty()) toks.push_back(cur); }
        if (toks.empty()) continue;
        size_t first_mod = 1;
        // (a piece that does not read as an operand is a modifier: "offset:16" behind "off", a bare "sc0")
        Opnd tmp;
        if (parse_operand(toks[0], tmp)) {
            if (in.no >= 6) { err = "too many operands"; return false; }
            in.o[in.no++] = tmp;
        }
        else first_mod = 0;
        for (size_t k = first_mod; k < toks.size(); k++) mods.push_back(toks[k]);
    }
    for (auto &m : mods) {
        const size_t c = m.find(':');
        const std::string key = c == std::string::npos ? m : m.substr(0, c), val = c == std::string::npos ? "" : m.substr(c + 1);
        auto num = [&](const std::string &v) -> i64 { return (i64)strtoll(v.c_str(), nullptr, 0); };
        if (key == "offset") in.off0 = (i32)num(val);
        else if (key == "offset0") in.off0 = (i32)num(val);
        else if (key == "offset1") in.off1 = (i32)num(val);
        else if (key == "sc0") in.ret = true;          // (on atomics: return the old value; on loads / stores: a cache policy)
        else if (key == "sc1") in.sc1 = true;
        else if (key == "nt") in.nt = true;
        else if (key == "glc" || key == "slc" || key == "dlc" || key == "gds" || key == "lds") {}
        else if (key == "row_mask") in.row_mask = (u8)num(val);
        else if (key == "bank_mask") in.bank_mask = (u8)num(val);
        else if (key == "bound_ctrl") in.bound_ctrl = true;
        else if (key == "fi") {}
        else if (key == "clamp") in.clamp = true;
        else if (key == "quad_perm") { unsigned a, b, cc, d; if (sscanf(val.c_str(), "[%u,%u,%u,%u]", &a, &b, &cc, &d) != 4) { err = "quad_perm"; return false; } in.dpp = (u16)(a | (b << 2) | (cc << 4) | (d << 6)); }
        else if (key == "row_shl") in.dpp = (u16)(0x100 + num(val));
        else if (key == "row_shr") in.dpp = (u16)(0x110 + num(val));
        else if (key == "row_ror") in.dpp = (u16)(0x120 + num(val));
        else if (key == "wave_shl") in.dpp = 0x130;
        else if (key == "wave_rol") in.dpp = 0x134;
        else if (key == "wave_shr") in.dpp = 0x138;
        else if (key == "wave_ror") in.dpp = 0x13C;
        else if (key == "row_mirror") in.dpp = 0x140;
        else if (key == "row_half_mirror") in.dpp = 0x141;
        else if (key == "row_bcast") in.dpp = num(val) == 15 ? 0x142 : 0x143;
        else if (key == "row_newbcast") in.dpp = (u16)(0x150 + num(val));
        else if (key == "dst_sel") in.dst_sel = parse_sel(val);
        else if (key == "src0_sel") in.src0_sel = parse_sel(val);
        else if (key == "src1_sel") in.src1_sel = parse_sel(val);
        else if (key == "dst_unused") in.dst_unused = val == "UNUSED_SEXT" ? UNUSED_SEXT : val == "UNUSED_PRESERVE" ? UNUSED_PRESERVE : UNUSED_PAD;
        else if (key == "bitop3") in.bitop3 = (u8)num(val);
        else if (key == "op_sel" || key == "op_sel_hi") {
            unsigned a = 0, b = 0, cc = 0; const int n = sscanf(val.c_str(), "[%u,%u,%u]", &a, &b, &cc);
            if (n < 2) { err = "op_sel"; return false; }
            (key == "op_sel" ? in.op_sel : in.op_sel_hi) = (u8)(a | (b << 1) | (cc << 2));
        }
        else if (key == "vmcnt" || key == "lgkmcnt" || key == "expcnt") {}
        else { err = "modifier " + m + " not handled"; return false; }
    }
    return true;
}

static std::string cache_dir()
{
    const char *e = getenv("GFX950EMU_CACHE");
    std::string d = e ? e : "/tmp/gfx950emu_cache";
    mkdir(d.c_str(), 0755);
    return d;
}

bool parse_text(CodeObject &co)
{
    if (co.parsed) return true;
    // a stable name for the disassembly of this code object: FNV-1a over the file
    u64 h = 1469598103934665603ull;
    for (u8 b : co.elf) { h ^= b; h *= 1099511628211ull; }
    char name[64]; snprintf(name, sizeof name, "/%016llx", (unsigned long long)h);
    const std::string base = cache_dir() + name, elfp = base + ".elf", txtp = base + ".s";
    if (access(txtp.c_str(), R_OK) != 0) {
        // (several processes may start on an empty cache at once: each writes files of its own and renames them into place)
        const std::string pid = std::to_string((long)getpid()), myelf = elfp + ".tmp." + pid, tmp = txtp + ".tmp." + pid;
        FILE *f = fopen(myelf.c_str(), "wb");
        if (!f) { fprintf(stderr, "gfx950emu: cannot write %s\n", myelf.c_str()); return false; }
        fwrite(co.elf.data(), 1, co.elf.size(), f); fclose(f);
        const char *od = getenv("GFX950EMU_OBJDUMP");
        const std::string cmd = std::string(od ? od : "/opt/rocm/lib/llvm/bin/llvm-objdump") + " -d --mcpu=gfx950 " + myelf + " > " + tmp + " 2>/dev/null && mv " + myelf + " " + elfp +
                                " && mv " + tmp + " " + txtp;
        if (system(cmd.c_str()) != 0 && access(txtp.c_str(), R_OK) != 0) { fprintf(stderr, "gfx950emu: %s failed\n", cmd.c_str()); return false; }
    }
    FILE *f = fopen(txtp.c_str(), "r");
    if (!f) return false;
    char *buf = nullptr; size_t cap = 0;
    u32 lineno = 0;
    std::string err;
    while (getline(&buf, &cap, f) > 0) {
        lineno++;
        if (buf[0] != '\t') continue;
        Inst in;
        if (!parse_line(buf, in, err)) {
            if (!err.empty()) { fprintf(stderr, "gfx950emu: %s line %u: %s: %s", txtp.c_str(), lineno, err.c_str(), buf); free(buf); fclose(f); return false; }
            continue;
        }
        in.line = lineno;
        co.at[in.addr] = (u32)co.insts.size();
        co.insts.push_back(in);
    }
    free(buf); fclose(f);
    // branch targets: address of the next instruction + 4 * simm16
    for (size_t i = 0; i < co.insts.size(); i++) {
        Inst &in = co.insts[i];
        if (in.op == OP_s_branch || (in.op >= OP_s_cbranch_scc0 && in.op <= OP_s_cbranch_execnz)) {
            const i64 t = (i64)in.addr + 4 + 4 * (i64)(i16)(u16)in.o[0].imm;
            auto it = co.at.find((u64)t);
            if (it == co.at.end()) { fprintf(stderr, "gfx950emu: branch at %llx to %llx: no instruction there\n", (unsigned long long)in.addr, (unsigned long long)t); return false; }
            in.target = it->second;
        }
    }
    for (auto &k : co.kernels) {
        auto it = co.at.find(k.second.entry);
        if (it == co.at.end()) { fprintf(stderr, "gfx950emu: kernel %s: no instruction at its entry\n", k.first.c_str()); return false; }
        k.second.first_inst = it->second;
    }
    co.parsed = true;
    return true;
}
