// wisdom.hpp -- what tuning runs have found, kept per process and exchanged as text (PlannerMode::Tune, planner.rs:18-32: the
// reference "benchmarks both paths at plan time and picks whichever is faster"; here the paths are the plans of plan.hpp and
// a result is worth keeping: a tuning run costs 0.1 .. 3 s, reading its answer nothing).
//
// One line per (type, call kind, log2 length, batch bucket):
//     f64 c2c 20 0 6,8,6@10,12,10:p8w fuse=0 us=23.10 heur=24.02
//     f32 r2c 24 0 heuristic fuse=1 us=88.0 heur=88.0          <- tuned, and the static rule's plan stood
// after a header "phastft-hip-wisdom 1 cus=<compute units of the device the times were taken on>".  Lengths are the CALLER's
// (the real length for r2c / c2r); a bucket b covers batches in (2^(b-1), 2^b] (0: one transform).  Entries measured on a
// device with another CU count are kept but never applied.
//
// Layers, later wins: built-in wisdom (builtin_wisdom.inc: generated on an MI355X by tools/make_builtin_wisdom.py; off with
// PHAST_BUILTIN_WISDOM=0) < the file named by PHAST_WISDOM (read on first use, rewritten after every tuning run) <
// phast_wisdom_import() < tuning runs of this process.
//
// No HIP in this header: the store is exercised without a GPU (tests/test_abi.py).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

struct WisdomEntry {
    bool heuristic = true;  // the static rule's plan was (one of) the fastest: nothing to install
    PlanSpec spec;
    bool fuse = false;  // r2c: the untangle rides in the last pass
    float us = 0, us_heur = 0;
    int cus = 0;        // compute units of the device it was measured on
    int layer = 0;      // 0 built-in, 1 file, 2 imported, 3 measured by this process
};

class WisdomStore {
  public:
    static WisdomStore &instance() {
        static WisdomStore *s = new WisdomStore();  // never destroyed: planners may outlive static destructors
        return *s;
    }
    static std::string key(size_t elem_bytes, int kind, unsigned log_n, unsigned bucket) {
        char b[64];
        std::snprintf(b, sizeof b, "%s %s %u %u", elem_bytes == 8 ? "f64" : "f32", kind_name(kind), log_n, bucket);
        return b;
    }
    bool lookup(size_t elem_bytes, int kind, unsigned log_n, unsigned bucket, int cus, WisdomEntry *out) {
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        auto it = entries_.find(key(elem_bytes, kind, log_n, bucket));
        if (it == entries_.end() || (it->second.cus && cus && it->second.cus != cus)) return false;
        if (out) *out = it->second;
        return true;
    }
    // every bucket known for (type, kind, length) on a device with `cus` compute units
    std::map<unsigned, WisdomEntry> lookup_all(size_t elem_bytes, int kind, unsigned log_n, int cus) {
        std::map<unsigned, WisdomEntry> out;
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        const std::string prefix = key(elem_bytes, kind, log_n, 0);
        const std::string stem = prefix.substr(0, prefix.size() - 1);  // "... <log_n> "
        for (auto it = entries_.lower_bound(stem); it != entries_.end() && it->first.compare(0, stem.size(), stem) == 0; ++it)
            if (!(it->second.cus && cus && it->second.cus != cus)) out[(unsigned)std::atoi(it->first.c_str() + stem.size())] = it->second;
        return out;
    }
    void record(size_t elem_bytes, int kind, unsigned log_n, unsigned bucket, const WisdomEntry &e) {
        std::string path;
        {
            std::lock_guard<std::mutex> lk(mu_);
            load_layers_locked();
            entries_[key(elem_bytes, kind, log_n, bucket)] = e;
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
        return export_locked(1);
    }
    size_t count(int layer) {  // entries of one layer (0 built-in .. 3 measured here), -1: all
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        size_t c = 0;
        for (const auto &kv : entries_) c += layer < 0 || kv.second.layer == layer;
        return c;
    }
    void forget() {  // everything but the built-in layer
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        for (auto it = entries_.begin(); it != entries_.end();) it = it->second.layer > 0 ? entries_.erase(it) : std::next(it);
    }
    // the built-in layer on or off at run time (tests that look at the static rules' plans; tools: A/B of the two)
    void set_builtin(bool on) {
        std::lock_guard<std::mutex> lk(mu_);
        load_layers_locked();
        for (auto it = entries_.begin(); it != entries_.end();) it = it->second.layer == 0 ? entries_.erase(it) : std::next(it);
        if (on) (void)import_locked(builtin_text(), 0, nullptr);  // (an entry of a later layer for the same key stays)
    }

  private:
    std::mutex mu_;
    std::map<std::string, WisdomEntry> entries_;
    bool loaded_ = false;
    std::string file_;

    static const char *builtin_text() {
        static const char text[] =
#include "builtin_wisdom.inc"
            ;
        return text;
    }
    void load_layers_locked() {
        if (loaded_) return;
        loaded_ = true;
        const char *off = std::getenv("PHAST_BUILTIN_WISDOM");
        if (!(off && *off == '0')) (void)import_locked(builtin_text(), 0, nullptr);
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
        if (!text) return -1;
        std::istringstream in(text);
        std::string line;
        int cus = 0;
        bool header = false;
        std::vector<std::pair<std::string, WisdomEntry>> parsed;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ls(line);
            std::string a;
            ls >> a;
            if (a == "phastft-hip-wisdom") {
                int version = 0;
                std::string c;
                ls >> version >> c;
                if (version != 1) return -1;
                cus = c.compare(0, 4, "cus=") == 0 ? std::atoi(c.c_str() + 4) : 0;
                header = true;
                continue;
            }
            if (!header) return -1;
            std::string kind, plan, tok;
            unsigned log_n = 0, bucket = 0;
            ls >> kind >> log_n >> bucket >> plan;
            const int k = kind_from_name(kind);
            WisdomEntry e;
            e.cus = cus;
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
        for (auto &kv : parsed) {
            auto it = entries_.find(kv.first);
            if (it == entries_.end() || it->second.layer <= layer) entries_[kv.first] = kv.second;
        }
        return 0;
    }
    std::string export_locked(int min_layer) const {
        // grouped by the CU count the entries were measured with (one header per group)
        std::map<int, std::string> groups;
        for (const auto &kv : entries_) {
            const WisdomEntry &e = kv.second;
            if (e.layer < min_layer) continue;
            char tail[96];
            std::snprintf(tail, sizeof tail, " fuse=%d us=%.2f heur=%.2f\n", e.fuse ? 1 : 0, (double)e.us, (double)e.us_heur);
            groups[e.cus] += kv.first + " " + (e.heuristic ? std::string("heuristic") : spec_to_string(e.spec)) + tail;
        }
        std::string out;
        for (const auto &g : groups) out += "phastft-hip-wisdom 1 cus=" + std::to_string(g.first) + "\n" + g.second;
        if (out.empty()) out = "phastft-hip-wisdom 1 cus=0\n";
        return out;
    }
    void save(const std::string &path) {
        std::string text;
        {
            std::lock_guard<std::mutex> lk(mu_);
            text = export_locked(1);
        }
        const std::string tmp = path + ".tmp";
        if (FILE *f = std::fopen(tmp.c_str(), "wb")) {
            const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
            if (std::fclose(f) == 0 && ok) (void)std::rename(tmp.c_str(), path.c_str());
            else (void)std::remove(tmp.c_str());
        }
    }
};

}  // namespace phast
