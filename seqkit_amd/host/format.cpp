// format.cpp -- text formatting that must match the Go command byte for byte: fmt "%.2f",
// go-humanize Comma/Commaf, util/math.Round, byteutil.WrapByteSlice, and the output sink.
#include <zlib.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "host.h"

namespace host {

int fatal(const std::string& msg)
{
    fprintf(stderr, "[ERRO] %s\n", msg.c_str());     // go-logging format "[%{level:.4s}] %{message}"
    return 255;                                       // os.Exit(-1)
}

Sink::Sink(const std::string& path) : buf_(new char[1 << 22])
{
    if (path == "-") { fp_ = stdout; ok_ = true; }
    else if (path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0) { gz_ = gzopen(path.c_str(), "wb"); ok_ = gz_ != nullptr; }
    else { fp_ = fopen(path.c_str(), "wb"); ok_ = fp_ != nullptr; }
}

Sink::~Sink()
{
    flush();
    if (gz_) gzclose((gzFile)gz_);
    else if (fp_ && fp_ != stdout) fclose(fp_);
    else if (fp_) fflush(fp_);
    delete[] buf_;
}

void Sink::flush()
{
    if (!len_) return;
    if (gz_) gzwrite((gzFile)gz_, buf_, (unsigned)len_);
    else if (fp_) fwrite(buf_, 1, len_, fp_);
    len_ = 0;
}

void Sink::write(const void* p, size_t n)
{
    if (n >= cap_) {
        flush();
        if (gz_) gzwrite((gzFile)gz_, p, (unsigned)n);
        else if (fp_) fwrite(p, 1, n, fp_);
        return;
    }
    if (len_ + n > cap_) flush();
    memcpy(buf_ + len_, p, n);
    len_ += n;
}

void Sink::puts(const char* s) { write(s, strlen(s)); }

// Go's %.Nf and C's agree digit for digit (both round the exact binary value); only the
// spelling of NaN / Inf differs.
void fmt_fixed(double v, int prec, std::string& out)
{
    if (std::isnan(v)) { out = "NaN"; return; }
    if (std::isinf(v)) { out = v > 0 ? "+Inf" : "-Inf"; return; }
    char buf[400];
    snprintf(buf, sizeof buf, "%.*f", prec, v);
    out = buf;
}

static std::string group3(const std::string& digits)
{
    std::string o;
    const size_t n = digits.size();
    for (size_t i = 0; i < n; i++) {
        if (i && (n - i) % 3 == 0) o.push_back(',');
        o.push_back(digits[i]);
    }
    return o;
}

std::string comma(int64_t v)
{
    const bool neg = v < 0;
    const uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
    return (neg ? "-" : "") + group3(std::to_string(u));
}

std::string commaf(double v)
{
    std::string sign;
    if (v < 0) { sign = "-"; v = -v; }
    char buf[64];
    for (int prec = 0; prec <= 20; prec++) {           // strconv.FormatFloat(v, 'f', -1, 64)
        snprintf(buf, sizeof buf, "%.*f", prec, v);
        if (strtod(buf, nullptr) == v) break;
    }
    std::string s(buf);
    const size_t dot = s.find('.');
    if (dot == std::string::npos) return sign + group3(s);
    return sign + group3(s.substr(0, dot)) + s.substr(dot);
}

double round_to(double f, int n)
{
    const double p = std::pow(10.0, n);
    return std::trunc((f + 0.5 / p) * p) / p;
}

void write_wrapped(Sink& out, const uint8_t* s, size_t n, int width)
{
    if (width < 1 || n == 0) { out.write(s, n); return; }
    const size_t w = (size_t)width;
    for (size_t i = 0; i < n; i += w) {
        if (i) out.put('\n');
        out.write(s + i, n - i < w ? n - i : w);
    }
}

void id_span(const GlobalOpts& g, const uint8_t* name, uint32_t n, uint32_t* off, uint32_t* len)
{
    if (!g.id_re) {
        // `^([^\s]+)\s?`: up to the first white space; no match (empty, or starts with white space): the whole line
        uint32_t i = 0;
        while (i < n && !(name[i] == ' ' || (name[i] >= '\t' && name[i] <= '\r' && name[i] != '\v'))) i++;
        *off = 0; *len = i == 0 ? n : i;
        return;
    }
    std::cmatch m;
    const char* b = reinterpret_cast<const char*>(name);
    if (std::regex_search(b, b + n, m, *g.id_re) && m.size() > 1 && m[1].matched) {
        *off = (uint32_t)(m[1].first - b); *len = (uint32_t)m[1].length();
    } else {
        *off = 0; *len = n;
    }
}

}  // namespace host
