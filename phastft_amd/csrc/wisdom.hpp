// wisdom.hpp -- what tuning runs have found, kept per process and exchanged as text (PlannerMode::Tune, planner.rs:18-32: the
// reference "benchmarks both paths at plan time and picks whichever is faster"; here the paths are the plans of plan.hpp and
// a result is worth keeping: a tuning run costs 0.1 .. 3 s, reading its answer nothing).
//
// One line per (type, call kind, log2 length, batch bucket):
//     f64 c2c 20 0 6,8,6@10,12,10:p8w fuse=0 us=23.10 heur=24.02
//     f32 r2c 24 0 heuristic fuse=1 us=88.0 heur=88.0          <- tuned, and the static rule's plan stood
// after a header "phastft-hip-wisdom 1 cus=<compute units> arch=<gfx name> lib=<plan generation>" naming the device the times
// were taken on and the generation of kernels that was timed (round 6; texts with "cus=" alone are read as before).  Lengths
// are the CALLER's (the real length for r2c / c2r); a bucket b covers batches in (2^(b-1), 2^b] (0: one transform).  Entries
// measured on a device with another CU count or architecture, or by another generation of the kernels, are kept (and exported)
// but never applied.
//
// Layers (kept apart, resolved at look-up), later wins: built-in wisdom (builtin_wisdom.inc: generated on an MI355X by tools/make_builtin_wisdom.py; off with
// PHAST_BUILTIN_WISDOM=0) < the file named by PHAST_WISDOM (read on first use, rewritten after every tuning run) <
// phast_wisdom_import() < tuning runs of this process.
//
// No HIP in this header: the store is exercised without a GPU (tests/test_abi.py).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "plan.hpp"

namespace phast {

enum CallKind : int { kC2C = 0, kC2CI = 1, kR2C = 2, kC2R = 3, kNumKinds = 4 };  // == PHAST_TUNE_* (phastft_hip.h)
inline const char *kind_name(int k) { return k == kC2C ? "c2c" : k == kC2CI ? "c2ci" : k == kR2C ? "r2c" : k == kC2R ? "c2r" : "?"; }
inline int kind_from_name(const std::string &s) {
    for (int k = 0; k < kNumKinds; ++k)
        if (s == kind_name(k)) return k;
    return -1;
}
// batches in (2^(b-1), 2^b] share a bucket; one transform is bucket 0
inline unsigned batch_bucket(size_t batch) { return batch <= 1 ? 0u : 64u - (unsigned)__builtin_clzll((unsigned long long)(batch - 1)); }

// What a line was measured with.  `arch` is the gfx number of the device as hex digits (gfx950 -> 0x950, gfx90a -> 0x90a),
// `lib` the plan generation of the library that timed it (kWisdomLib: bumped whenever a pass kernel changes enough that old
// rankings say nothing about the new code).  0 = unknown (texts written before round 6 carry neither): matches anything.
constexpr int kWisdomLib = 6;
struct WisdomOrigin {
    int cus = 0, arch = 0, lib = 0;
    bool operator<(const WisdomOrigin &o) const { return cus != o.cus ? cus < o.cus : arch != o.arch ? arch < o.arch : lib < o.lib; }
};
inline int arch_from_name(const char *gcn_arch_name) {  // "gfx950:sramecc+:xnack-" -> 0x950; anything else -> 0
    if (!gcn_arch_name || std::strncmp(gcn_arch_name, "gfx", 3) != 0) return 0;
    char *end = nullptr;
    const long v = std::strtol(gcn_arch_name + 3, &end, 16);
    return (end == gcn_arch_name + 3 || v <= 0 || v > 0xfffff) ? 0 : (int)v;
}

struct WisdomEntry {
    bool heuristic = true;  // the static rule's plan was (one of) the fastest: nothing to install
    PlanSpec spec;
    bool fuse = false;  // r2c: the untangle rides in the last pass
    float us = 0, us_heur = 0;
    int cus = 0;        // compute units of the device it was measured on
    int arch = 0;       // ... its gfx number (arch_from_name), 0 = not recorded
    int lib = 0;        // ... and the library's plan generation (kWisdomLib), 0 = not recorded
    int layer = 0;      // 0 built-in, 1 file, 2 imported, 3 measured by this process
    // applied only on the kind of device, and by the generation of kernels, it was measured with
    bool applies(int dev_cus, int dev_arch) const {
        return !(cus && dev_cus && cus != dev_cus) && !(arch && dev_arch && arch != dev_arch) && !(lib && lib != kWisdomLib);
    }
};

class WisdomStore {
  public:
    static constexpr int kLayers = 4;
    static WisdomStore &instance() {
        static WisdomStore *s = new WisdomStore();  // never destroyed: planners may outlive static destructors
        return *s;
    }
    static std::string key(size_t elem_bytes, int kind, unsigned log_n, unsigned bucket) {
        char b[64];
        std::snprintf(b, sizeof b, "%s %s %u %u", elem_bytes == 8 ? "f64" : "f32", kind_name(kind), log_n, bucket);
        return b;
    }
    bool lookup(size_t elem_bytes, int kind, unsigned log_n, unsigned bucket, int cus, int arch, WisdomEntry *out) {
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        const WisdomEntry *e = find_locked(key(elem_bytes, kind, log_n, bucket));
        if (!e || !e->applies(cus, arch)) return false;
        if (out) *out = *e;
        return true;
    }
    // every bucket known for (type, kind, length) on a device with `cus` compute units of architecture `arch`
    std::map<unsigned, WisdomEntry> lookup_all(size_t elem_bytes, int kind, unsigned log_n, int cus, int arch) {
        std::map<unsigned, WisdomEntry> out;
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        const std::string prefix = key(elem_bytes, kind, log_n, 0);
        const std::string stem = prefix.substr(0, prefix.size() - 1);  // "... <log_n> "
        for (int layer = builtin_on_ ? 0 : 1; layer < kLayers; ++layer)  // later layers overwrite earlier ones
            for (auto it = layers_[layer].lower_bound(stem); it != layers_[layer].end() && it->first.compare(0, stem.size(), stem) == 0; ++it)
                out[(unsigned)std::atoi(it->first.c_str() + stem.size())] = it->second;
        for (auto it = out.begin(); it != out.end();) it = it->second.applies(cus, arch) ? std::next(it) : out.erase(it);
        return out;
    }
    void record(size_t elem_bytes, int kind, unsigned log_n, unsigned bucket, const WisdomEntry &e) {
        std::string path;
        {
            std::lock_guard<std::mutex> lk(mu_);
            load_layers_locked();
            const int layer = e.layer >= 0 && e.layer < kLayers ? e.layer : 3;
            layers_[layer][key(elem_bytes, kind, log_n, bucket)] = e;
            path = file_;
        }
        if (!path.empty()) save(path);
    }
    // 0 = ok, -1 = not a wisdom text.  Lines that do not parse are skipped and counted in *skipped.
    int import_text(const char *text, int layer, size_t *skipped = nullptr) {
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        return import_locked(text, layer, skipped);
    }
    // what this process measured, imported or read from PHAST_WISDOM -- NOT the built-in layer: a text that carried it would pin
    // this library version's built-in plans in every file and process it travels to
    std::string export_text() {
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        return export_locked();
    }
    size_t count(int layer) {  // keys whose winning entry belongs to `layer` (0 built-in .. 3 measured here), -1: all keys
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        size_t c = 0;
        for (int l = builtin_on_ ? 0 : 1; l < kLayers; ++l)
            for (const auto &kv : layers_[l]) {
                bool shadowed = false;
                for (int h = l + 1; h < kLayers && !shadowed; ++h) shadowed = layers_[h].count(kv.first) != 0;
                c += !shadowed && (layer < 0 || layer == l);
            }
        return c;
    }
    void forget() {  // everything but the built-in layer (which is whole again afterwards: the layers are kept apart)
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        for (int l = 1; l < kLayers; ++l) layers_[l].clear();
    }
    // the built-in layer on or off at run time (tests that look at the static rules' plans; tools: A/B of the two).  Returns
    // what it was, so that a caller can put it back (PHAST_BUILTIN_WISDOM=0 is "off" from the start).
    bool set_builtin(bool on) {
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        const bool was = builtin_on_;
        builtin_on_ = on;
        return was;
    }

  private:
    std::mutex mu_, save_mu_;
    std::map<std::string, WisdomEntry> layers_[kLayers];
    bool loaded_ = false, builtin_on_ = true;
    std::string file_;

    static const char *builtin_text() {
        static const char text[] =
#include "builtin_wisdom.inc"
            ;
        return text;
    }
    const WisdomEntry *find_locked(const std::string &k) const {
        for (int l = kLayers - 1; l >= (builtin_on_ ? 0 : 1); --l) {
            auto it = layers_[l].find(k);
            if (it != layers_[l].end()) return &it->second;
        }
        return nullptr;
    }
    void load_layers_locked() {
        if (loaded_) return;
        loaded_ = true;
        (void)import_locked(builtin_text(), 0, nullptr);
        const char *off = std::getenv("PHAST_BUILTIN_WISDOM");
        builtin_on_ = !(off && *off == '0');
        const char *path = std::getenv("PHAST_WISDOM");
        if (path && *path) {
            file_ = path;
            if (FILE *f = std::fopen(path, "rb")) {
                std::string text;
                char buf[4096];
                size_t got;
                while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
                std::fclose(f);
                (void)import_locked(text.c_str(), 1, nullptr);
            }
        }
    }
    // all or nothing: a text that is refused (no header, another format version -- also half way down) leaves the store as it was
    int import_locked(const char *text, int layer, size_t *skipped) {
        if (skipped) *skipped = 0;
        if (!text || layer < 0 || layer >= kLayers) return -1;
        std::istringstream in(text);
        std::string line;
        WisdomOrigin org;
        bool header = false;
        std::vector<std::pair<std::string, WisdomEntry>> parsed;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ls(line);
            std::string a;
            ls >> a;
            if (a == "phastft-hip-wisdom") {  // "phastft-hip-wisdom 1 cus=256 [arch=gfx950] [lib=6]"
                int version = 0;
                std::string c;
                ls >> version;
                if (version != 1) return -1;
                org = WisdomOrigin();
                while (ls >> c) {
                    if (c.compare(0, 4, "cus=") == 0) org.cus = std::atoi(c.c_str() + 4);
                    else if (c.compare(0, 5, "arch=") == 0) org.arch = arch_from_name(c.c_str() + 5);
                    else if (c.compare(0, 4, "lib=") == 0) org.lib = std::atoi(c.c_str() + 4);
                }
                if (org.cus < 0) org.cus = 0;
                if (org.lib < 0) org.lib = 0;
                header = true;
                continue;
            }
            if (!header) return -1;
            std::string kind, plan, tok;
            unsigned log_n = 0, bucket = 0;
            ls >> kind >> log_n >> bucket >> plan;
            const int k = kind_from_name(kind);
            WisdomEntry e;
            e.cus = org.cus;
            e.arch = org.arch;
            e.lib = org.lib;
            e.layer = layer;
            bool ok = (a == "f64" || a == "f32") && k >= 0 && !ls.fail() && log_n >= 1 && log_n <= 40 && bucket <= 40;
            if (ok) {
                e.heuristic = plan == "heuristic";
                if (!e.heuristic) ok = spec_from_string(plan.c_str(), e.spec);
            }
            while (ok && (ls >> tok)) {
                if (tok.compare(0, 5, "fuse=") == 0) e.fuse = tok[5] == '1';
                else if (tok.compare(0, 3, "us=") == 0) e.us = (float)std::atof(tok.c_str() + 3);
                else if (tok.compare(0, 5, "heur=") == 0) e.us_heur = (float)std::atof(tok.c_str() + 5);
            }
            if (!ok) {
                if (skipped) ++*skipped;
                continue;
            }
            parsed.emplace_back(key(a == "f64" ? 8 : 4, k, log_n, bucket), e);
        }
        if (!header) return -1;
        for (auto &kv : parsed) layers_[layer][kv.first] = kv.second;
        return 0;
    }
    std::string export_locked() const {
        // the winning entry of every key outside the built-in layer, grouped by what the entries were measured with (one header
        // per group; arch= / lib= only where recorded, so that a text from before round 6 comes out as it went in)
        std::map<std::string, const WisdomEntry *> win;
        for (int l = 1; l < kLayers; ++l)
            for (const auto &kv : layers_[l]) win[kv.first] = &kv.second;
        std::map<WisdomOrigin, std::string> groups;
        for (const auto &kv : win) {
            const WisdomEntry &e = *kv.second;
            char tail[96];
            std::snprintf(tail, sizeof tail, " fuse=%d us=%.2f heur=%.2f\n", e.fuse ? 1 : 0, (double)e.us, (double)e.us_heur);
            WisdomOrigin o;
            o.cus = e.cus;
            o.arch = e.arch;
            o.lib = e.lib;
            groups[o] += kv.first + " " + (e.heuristic ? std::string("heuristic") : spec_to_string(e.spec)) + tail;
        }
        std::string out;
        for (const auto &g : groups) {
            char head[96];
            int at = std::snprintf(head, sizeof head, "phastft-hip-wisdom 1 cus=%d", g.first.cus);
            if (g.first.arch) at += std::snprintf(head + at, sizeof head - (size_t)at, " arch=gfx%x", g.first.arch);
            if (g.first.lib) at += std::snprintf(head + at, sizeof head - (size_t)at, " lib=%d", g.first.lib);
            out += std::string(head) + "\n" + g.second;
        }
        if (out.empty()) out = "phastft-hip-wisdom 1 cus=0\n";
        return out;
    }
    void save(const std::string &path) {
        // one writer at a time and a temporary name of its own: two tuning runs finishing together (threads of this process, or
        // two processes sharing PHAST_WISDOM) never interleave their writes -- the file is always one whole export
        std::lock_guard<std::mutex> sl(save_mu_);
        std::string text;
        {
            std::lock_guard<std::mutex> lk(mu_);
            text = export_locked();
        }
        static unsigned long counter = 0;
        char suffix[64];
        std::snprintf(suffix, sizeof suffix, ".tmp.%ld.%lu", (long)::getpid(), ++counter);
        const std::string tmp = path + suffix;
        if (FILE *f = std::fopen(tmp.c_str(), "wb")) {
            const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
            if (std::fclose(f) == 0 && ok) (void)std::rename(tmp.c_str(), path.c_str());
            else (void)std::remove(tmp.c_str());
        }
    }
};

}  // namespace phast
